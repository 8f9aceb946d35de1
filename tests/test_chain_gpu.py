"""The chained tail of an SVI step (csrc/chain.hip) and the hoisted constants of a captured step
(infer/constants.py) on the MI355X.

The chain runs the SAME device code with the same thread geometry as the four stand-alone launches
it replaces, so every comparison here is BITWISE: loss values and parameters of chained steps
against unchained steps from the same seed."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _clean():
    import pyro_amd
    pyro_amd.clear_param_store()
    yield
    pyro_amd.clear_param_store()
    pyro_amd.enable_validation(True)


def _setup(gpu, N=20000, D=32, P=64, seed=11):
    import pyro_amd as pyro
    from pyro_amd import examples
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    X, y = examples.synthetic_logreg_data(N, D, gpu, seed=3)
    pyro.clear_param_store()
    pyro.set_rng_seed(seed)
    pyro.enable_validation(False)
    guide = AutoNormal(examples.logreg_model, init_scale=0.1)
    svi = SVI(examples.logreg_model, guide, pyro.optim.Adam({"lr": 0.02}),
              Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1))
    return pyro, svi, X, y


def _params(pyro):
    return {k: v.detach().clone() for k, v in pyro.get_param_store().items()}


def _chained_eager_step(svi, X, y):
    """One eager step whose tail is recorded and launched as a chain."""
    from pyro_amd import kernels, poutine
    with kernels.chain_recording(X.device) as rec:
        with poutine.trace(param_only=True) as pc:
            loss = svi._loss_device(svi.model, svi.guide, X, y)
        params = svi._params_of(pc)
        svi.optim(params)
    return float(loss), rec.stats + (rec.fused,)


@pytest.mark.parametrize("P", [64, 16])
def test_chained_tail_is_bitwise_the_separate_launches(gpu, P):
    """P = 64: plane-image GLM kernel (after the second sighting) ; P = 16: the bf16x3 kernel.
    Unchained / chained phase by phase / chained with the per-site fused tail."""
    from pyro_amd import kernels
    runs = []
    try:
        for chained in (False, "phases", "fused", "fast"):
            kernels.chain_tune(fuse_tail=chained in ("fused", "fast"), fast_sites=chained == "fast")
            pyro, svi, X, y = _setup(gpu, P=P)
            losses, stats = [], None
            for i in range(6):
                if chained and i >= 2:        # (the first steps create parameters / optimizer state)
                    loss, stats = _chained_eager_step(svi, X, y)
                else:
                    loss = svi.step(X, y)
                losses.append(loss)
            runs.append((losses, _params(pyro), stats))
    finally:
        kernels.chain_tune(fuse_tail=True)
    assert runs[1][2] == (1, 4, 0), runs[1][2]        # ONE launch carrying all four phases
    assert runs[2][2] == (1, 4, 1), runs[2][2]        # ... in the fused form
    assert runs[3][2] == (1, 4, 1), runs[3][2]        # ... with the one-pass site code
    for other in runs[1:]:
        assert runs[0][0] == other[0], (runs[0][0], other[0])
        for k in runs[0][1]:
            assert torch.equal(runs[0][1][k], other[1][k]), k


def test_captured_step_is_three_nodes_of_ours(gpu):
    """SVI(hip_graph=True) on the config-2 model text: the capture holds the GLM kernel -- which makes
    the guide's draw in its own prologue (round 4; until then the draw was a node of its own) -- and
    ONE chain launch; the two ``X.new_zeros`` of the model are served from hoisted constants; the
    trajectory is bitwise the eager one (whose draw IS a launch of its own)."""
    import pyro_amd as pyro
    from pyro_amd import examples
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    out = []
    for graph in (False, True):
        # fresh tensors per run: the plane image of a design matrix is cached per tensor object and
        # decides which GLM kernel the FIRST steps run
        X, y = examples.synthetic_logreg_data(20000, 32, gpu, seed=3)
        pyro.clear_param_store()
        pyro.set_rng_seed(5)
        pyro.enable_validation(False)
        guide = AutoNormal(examples.logreg_model, init_scale=0.1)
        svi = SVI(examples.logreg_model, guide, pyro.optim.Adam({"lr": 0.02}),
                  Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1),
                  hip_graph=graph, graph_warmup=3)
        losses = [svi.step(X, y) for _ in range(10)]
        if graph:
            assert svi.hip_graph and len(svi._graphs) == 1
            assert svi.chain_stats == [(1, 4)] and svi.chain_fused == 1, svi.chain_stats
            entry = next(iter(svi._graphs.values()))
            assert entry.constants_served == 2, entry.constants_served
        out.append((losses, _params(pyro)))
    assert out[0][0] == out[1][0]
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k


def test_torch_operator_between_phases_flushes_the_chain(gpu):
    """A torch kernel that reads what a pending phase writes must see it written: the dispatch
    guard launches the pending phases first."""
    from pyro_amd import kernels

    g = torch.Generator(device=gpu).manual_seed(0)
    N, D, P = 4096, 32, 8
    X = torch.randn((N, D), device=gpu, generator=g)
    y = (torch.rand((N,), device=gpu, generator=g) < 0.5).float()
    w = torch.randn((P, D), device=gpu, generator=g) * 0.1
    b = torch.randn((P,), device=gpu, generator=g) * 0.1
    ll0, gw0, gb0 = kernels.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0)
    with kernels.chain_recording(gpu) as rec:
        ll, gw, gb = kernels.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0)   # finalize is pending
        from pyro_amd import _lib
        assert _lib.load().pa_chain_pending() == 1
        twice = ll * 2.0                                                    # a torch kernel reads ll
        assert _lib.load().pa_chain_pending() == 0
    assert rec.stats == (1, 1)
    assert torch.equal(ll, ll0) and torch.equal(gw, gw0) and torch.equal(gb, gb0)
    assert torch.equal(twice, ll0 * 2.0)


def test_a_step_that_writes_into_a_fresh_constant_is_captured_with_its_fills(gpu):
    """``z = X.new_zeros(D); z += 1`` must be re-filled on every replay: the capture with hoisted
    constants is refused and redone with the fill nodes inside the graph."""
    import pyro_amd as pyro
    from pyro_amd import distributions as dist
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    X = torch.randn((512, 4), device=gpu)
    yv = (torch.rand((512,), device=gpu) < 0.5).float()

    def model(X, y):
        loc = X.new_zeros(4)
        loc += 0.5                                    # in place on the fresh tensor
        w = pyro.sample("w", dist.Normal(loc, 1.0).to_event(1))
        with pyro.plate("data", X.shape[0]):
            pyro.sample("obs", dist.Bernoulli(logits=(X * w.unsqueeze(-2)).sum(-1)), obs=y)

    out = []
    for graph in (False, True):
        pyro.clear_param_store()
        pyro.set_rng_seed(2)
        pyro.enable_validation(False)
        svi = SVI(model, AutoNormal(model), pyro.optim.Adam({"lr": 0.05}),
                  Trace_ELBO(num_particles=8, vectorize_particles=True, max_plate_nesting=1),
                  hip_graph=graph, graph_warmup=2)
        losses = [svi.step(X, yv) for _ in range(8)]
        if graph:
            assert svi.hip_graph and len(svi._graphs) == 1
            assert next(iter(svi._graphs.values())).constants_served == 0
        out.append(losses)
    assert out[0] == out[1]


def test_in_kernel_finalize_is_bitwise_the_separate_launch(gpu):
    """pa_glm_bernoulli_planes_fwd_bwd: the two-level last-arriver reduction inside the kernel against
    the stand-alone finalize kernel, incl. a plate that does not fill the grid and P > 64 (two
    particle passes)."""
    from pyro_amd import kernels
    g = torch.Generator(device=gpu).manual_seed(0)
    try:
        for N, D, P in ((200_000, 32, 64), (1000, 17, 40), (70_000, 32, 130), (64, 8, 33)):
            X = torch.randn((N, D), device=gpu, generator=g)
            y = (torch.rand((N,), device=gpu, generator=g) < 0.5).float()
            w = torch.randn((P, D), device=gpu, generator=g) * 0.1
            b = torch.randn((P,), device=gpu, generator=g) * 0.1
            planes = kernels.glm_pack_planes(X)
            outs = []
            for mode in (False, True, True):
                kernels.glm_planes_finalize_mode(mode)
                outs.append(kernels.glm_bernoulli_planes_fwd_bwd(planes, y, w, b, 1.0, N, D))
            for o in outs[1:]:
                for a, r in zip(o, outs[0]):
                    assert torch.equal(a, r), (N, D, P)
    finally:
        kernels.glm_planes_finalize_mode(False)


def test_chain_can_be_switched_off(gpu, monkeypatch):
    monkeypatch.setenv("PYRO_AMD_CHAIN", "0")
    monkeypatch.setenv("PYRO_AMD_HOIST", "0")
    import pyro_amd as pyro
    pyro, svi, X, y = _setup(gpu)
    svi.hip_graph, svi.graph_warmup = True, 2
    losses = [svi.step(X, y) for _ in range(5)]
    assert svi.hip_graph and svi.chain_stats == [None]
    assert all(l == l for l in losses)


# ---- the step gate: replays enqueued ahead of the host (SVI(prearm=True), pa_gate) ----------------
def _gate_run(gpu, prearm, between=None, steps=24, seed=7, N=20000, speculate=True, D=32):
    import pyro_amd as pyro
    from pyro_amd import examples
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    X, y = examples.synthetic_logreg_data(N, D, gpu, seed=3)
    pyro.clear_param_store()
    pyro.set_rng_seed(seed)
    pyro.enable_validation(False)
    guide = AutoNormal(examples.logreg_model, init_scale=0.1)
    svi = SVI(examples.logreg_model, guide, pyro.optim.Adam({"lr": 0.02}),
              Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1),
              hip_graph=True, graph_warmup=3, prearm=prearm, speculate=speculate)
    losses = []
    for i in range(steps):
        if between is not None:
            between(i, svi, X, y)
        losses.append(svi.step(X, y))
    torch.cuda.synchronize()
    return losses, _params(pyro), svi


@pytest.mark.parametrize("speculate", [False, True], ids=["gate_first", "gate_before_tail"])
def test_prearmed_steps_are_bitwise_the_ordinary_ones(gpu, speculate):
    """Every node of the config-2 step behind the gate polls it: the capture is armable, from the second
    replay on each step() finds its replay already enqueued, and the trajectory is bit-identical -- with
    the gate as the first node, and with the gate in front of the chained tail (the GLM kernel of the
    replay enqueued ahead then runs before the host has asked for the step)."""
    l0, p0, _ = _gate_run(gpu, prearm=False)
    l1, p1, svi = _gate_run(gpu, prearm=True, speculate=speculate)
    (entry,) = svi._graphs.values()
    g = entry.gate
    assert g is not None and g.armable and g.late == speculate and g.torch_ops == 0
    if speculate:
        # in front of the gate: the GLM kernel (which draws the guide's sample itself); behind: the gate
        # node and the chained tail
        assert (g.pre, g.pre_other, g.total, g.aware) == (1, 0, 2, 2), (g.pre, g.pre_other, g.total, g.aware)
    else:
        # gate node, the GLM kernel, the chained tail
        assert (g.total, g.aware) == (3, 3), (g.total, g.aware)
    assert entry.armed and entry.arm_backoff == 0
    assert g.next == 24 - 3 + 1                     # every replay ran exactly once
    assert l0 == l1
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k
    # the replay left armed at the end gives itself up: the stream drains (synchronize returns) and
    # nothing has changed
    torch.cuda.synchronize()
    import pyro_amd as pyro
    for k, v in _params(pyro).items():
        assert torch.equal(v, p1[k]), k
    assert int(g.ack_np[0]) == g.next


@pytest.mark.parametrize("speculate", [False, True], ids=["gate_first", "gate_before_tail"])
def test_pause_gives_the_waiting_replay_up_at_once_and_arming_goes_on(gpu, speculate):
    """SVI.pause() before a synchronisation: the armed replay returns without its 40 us of patience
    (nothing has changed), later steps arm again, and the trajectory is the un-armed one bit for bit."""
    import time

    waits = []

    def pause_and_sync(i, svi, X, y):
        if i in (6, 7, 13, 20):
            svi.pause()
            (entry,) = svi._graphs.values()
            assert not entry.armed
            t0 = time.perf_counter()
            torch.cuda.synchronize()
            waits.append(time.perf_counter() - t0)

    l0, p0, _ = _gate_run(gpu, prearm=False)
    l1, p1, svi = _gate_run(gpu, prearm=True, between=pause_and_sync, speculate=speculate)
    assert l0 == l1
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k
    (entry,) = svi._graphs.values()
    assert entry.gate.next == 24 - 3 + 1 and entry.armed and svi.prearm
    assert len(waits) == 4 and max(waits) < 5e-3


@pytest.mark.parametrize("speculate", [False, True], ids=["gate_first", "gate_before_tail"])
def test_gate_gives_a_replay_up_when_the_host_stays_away(gpu, speculate):
    """The host sleeps between steps (longer than the gate's 40 us): the armed replay has given
    itself up, the step runs the ordinary way, arming backs off -- same trajectory."""
    import time

    def nap(i, svi, X, y):
        if i in (8, 9, 15):
            time.sleep(0.003)

    l0, p0, _ = _gate_run(gpu, prearm=False)
    l1, p1, svi = _gate_run(gpu, prearm=True, between=nap, speculate=speculate)
    assert l0 == l1
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k
    (entry,) = svi._graphs.values()
    assert entry.gate.next == 24 - 3 + 1


@pytest.mark.parametrize("speculate", [False, True], ids=["gate_first", "gate_before_tail"])
def test_writes_between_steps_cancel_the_armed_replay(gpu, speculate):
    """In-place writes to an argument tensor or to a parameter between two steps are enqueued BEHIND
    the armed replay; step() notices the version counters and cancels it, so the step sees them."""
    import pyro_amd as pyro

    def meddle(i, svi, X, y):
        if i == 10:
            X.mul_(0.5)                                   # new data in the same tensor
        if i == 14:
            with torch.no_grad():
                pyro.get_param_store()._params["AutoNormal.locs.w"].mul_(0.9)
        if i == 18:
            pyro.set_rng_seed(123)                        # the host-side stream position moved

    l0, p0, _ = _gate_run(gpu, prearm=False, between=meddle)
    l1, p1, svi = _gate_run(gpu, prearm=True, between=meddle, speculate=speculate)
    (entry,) = svi._graphs.values()
    assert entry.gate is not None and entry.gate.late == speculate
    assert l0 == l1
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k


def test_a_step_with_more_than_the_glm_kernel_before_its_tail_keeps_the_gate_first(gpu):
    """D = 64: the guide draw is a launch of its own in front of the feature-tile GLM kernel, so the gate
    may not move behind them (a late gate lets only the plane-image GLM kernel run ahead): the capture
    falls back to the gate as first node, or to none -- same trajectory either way."""
    l0, p0, _ = _gate_run(gpu, prearm=False, D=64)
    l1, p1, svi = _gate_run(gpu, prearm=True, speculate=True, D=64)
    (entry,) = svi._graphs.values()
    assert entry.gate is None or (entry.gate.armable and not entry.gate.late)
    assert l0 == l1
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k


def test_a_step_with_a_torch_kernel_in_it_is_captured_without_a_gate(gpu):
    """prearm=True on a step that launches something which cannot be given up (here: the model's
    logits are materialised by torch operators): no gate node, ordinary replays."""
    import pyro_amd as pyro
    from pyro_amd import distributions as dist
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    X = torch.randn((2000, 4), device=gpu)
    yv = (torch.rand((2000,), device=gpu) < 0.5).float()

    def model(X, y):
        w = pyro.sample("w", dist.Normal(X.new_zeros(4), 1.0).to_event(1))
        with pyro.plate("data", X.shape[0]):
            pyro.sample("obs", dist.Bernoulli(logits=torch.tanh(X * w.unsqueeze(-2)).sum(-1)), obs=y)

    out = []
    for prearm in (False, True):
        pyro.clear_param_store()
        pyro.set_rng_seed(2)
        pyro.enable_validation(False)
        svi = SVI(model, AutoNormal(model), pyro.optim.Adam({"lr": 0.05}),
                  Trace_ELBO(num_particles=8, vectorize_particles=True, max_plate_nesting=1),
                  hip_graph=True, graph_warmup=2, prearm=prearm)
        out.append([svi.step(X, yv) for _ in range(8)])
        (entry,) = svi._graphs.values()
        assert entry.gate is None
    assert out[0] == out[1]


# ---- race hunting: the device-wide arrival counters under shuffled timing --------------------------
@pytest.mark.parametrize("form", ["fused", "phases"])
def test_chain_kernels_are_bitwise_stable_under_shuffled_arrivals(gpu, form):
    """SURVEY section 5 (race detection): the chained tail orders its phases with device-wide arrival
    counters (one release fence per workgroup, relaxed spin + one acquire) and the captured step hands
    its loss over through a pinned mailbox.  With ``jitter_seed`` every workgroup sleeps a pseudo-random
    0..17 us in front of each arrival and after each wait, so that across seeds every workgroup gets to
    be first and last at every counter.  A missing fence or a wait on the wrong count shows up as a
    changed bit: 16 seeds x 6 chained steps (eager, chained phase by phase or in the per-site fused
    form) and 3 seeds x 40 replays of the captured step against the un-chained trajectory."""
    from pyro_amd import kernels
    fuse = form == "fused"
    try:
        kernels.chain_tune(fuse_tail=fuse)
        pyro, svi, X, y = _setup(gpu, P=64)
        ref_losses, ref_params = [], None
        for i in range(8):
            ref_losses.append(svi.step(X, y))                     # separate launches
        ref_params = _params(pyro)
        for seed in range(1, 17):
            kernels.chain_tune(fuse_tail=fuse, jitter_seed=seed * 7919)
            pyro, svi, X, y = _setup(gpu, P=64)
            losses = []
            for i in range(8):
                losses.append(_chained_eager_step(svi, X, y)[0] if i >= 2 else svi.step(X, y))
            assert losses == ref_losses, (seed, losses, ref_losses)
            got = _params(pyro)
            for k in ref_params:
                assert torch.equal(ref_params[k], got[k]), (seed, k)
    finally:
        kernels.chain_tune(fuse_tail=True)


def test_captured_step_is_bitwise_stable_under_shuffled_arrivals(gpu):
    import pyro_amd as pyro
    from pyro_amd import examples, kernels
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    def run(graph, seed, prearm=False, steps=43):
        kernels.chain_tune(fuse_tail=True, jitter_seed=seed)
        X, y = examples.synthetic_logreg_data(20000, 32, gpu, seed=3)
        pyro.clear_param_store()
        pyro.set_rng_seed(5)
        pyro.enable_validation(False)
        guide = AutoNormal(examples.logreg_model, init_scale=0.1)
        svi = SVI(examples.logreg_model, guide, pyro.optim.Adam({"lr": 0.02}),
                  Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1),
                  hip_graph=graph, graph_warmup=3, prearm=prearm)
        losses = [svi.step(X, y) for _ in range(steps)]
        torch.cuda.synchronize()
        return losses, _params(pyro)

    def step_ms(seed):
        """Mean time of one replay of the captured step (the delays must be real for the test to mean
        anything; they are launch arguments of the chain kernel, fixed at capture)."""
        import time
        kernels.chain_tune(fuse_tail=True, jitter_seed=seed)
        X, y = examples.synthetic_logreg_data(20000, 32, gpu, seed=3)
        pyro.clear_param_store()
        pyro.set_rng_seed(5)
        pyro.enable_validation(False)
        guide = AutoNormal(examples.logreg_model, init_scale=0.1)
        svi = SVI(examples.logreg_model, guide, pyro.optim.Adam({"lr": 0.02}),
                  Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1),
                  hip_graph=True, graph_warmup=3)
        for _ in range(10):
            svi.step(X, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            svi.step(X, y)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 200 * 1e3

    try:
        plain, shuffled = step_ms(0), step_ms(4242)
        assert shuffled > plain + 0.004, (plain, shuffled)       # >= 4 us of injected delay per step
        ref = run(False, 0)
        for seed, prearm in ((11, False), (977, False), (31337, True)):
            got = run(True, seed, prearm)
            assert got[0] == ref[0], seed
            for k in ref[1]:
                assert torch.equal(ref[1][k], got[1][k]), (seed, k)
    finally:
        kernels.chain_tune(fuse_tail=True)


def test_guide_draw_made_by_the_glm_kernel_is_bitwise_the_separate_launch(gpu):
    """Inside a recording pa_meanfield_normal_sample parks its launch; the plane-image GLM kernel whose
    weights and bias are two of its sites' draws makes them in its prologue (same Philox blocks, same
    softplus, one fma) and workgroup 0 stores z / eps / scale / loc for the tail.  Against the draw as
    its own launch (chain_tune(fuse_draw=False)): identical losses and parameters, with and without a
    bias site, for D a multiple of 4 (two Philox blocks per thread) and not (element by element)."""
    import pyro_amd as pyro
    from pyro_amd import distributions as dist, kernels
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    def run(D, with_bias, fuse_draw):
        kernels.chain_tune(fuse_tail=True, fuse_draw=fuse_draw)
        g = torch.Generator().manual_seed(D)
        X = torch.randn((9000, D), generator=g).to(gpu)
        yv = (torch.rand((9000,), generator=g) < 0.5).float().to(gpu)

        def model(X, y):
            w = pyro.sample("w", dist.Normal(X.new_zeros(D), 1.0).to_event(1))
            extra = pyro.sample("extra", dist.Normal(X.new_zeros(3), 2.0).to_event(1))   # a site the kernel does not draw
            logits = w @ X.t()
            if with_bias:
                logits = logits + pyro.sample("b", dist.Normal(X.new_zeros(()), 1.0)).unsqueeze(-1)
            with pyro.plate("data", X.shape[0]):
                pyro.sample("obs", dist.Bernoulli(logits=logits.squeeze(-2) if logits.dim() > 2 else logits), obs=y)
            return extra

        pyro.clear_param_store()
        pyro.set_rng_seed(4)
        pyro.enable_validation(False)
        svi = SVI(model, AutoNormal(model, init_scale=0.1), pyro.optim.Adam({"lr": 0.03}),
                  Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1),
                  hip_graph=True, graph_warmup=3)
        losses = [svi.step(X, yv) for _ in range(9)]
        return losses, _params(pyro)

    try:
        for D, with_bias in ((32, True), (20, True), (13, True), (32, False)):
            a, b = run(D, with_bias, False), run(D, with_bias, True)
            assert a[0] == b[0], (D, with_bias, a[0], b[0])
            for k in a[1]:
                assert torch.equal(a[1][k], b[1][k]), (D, with_bias, k)
    finally:
        kernels.chain_tune(fuse_tail=True)


@pytest.mark.parametrize("D", [64, 100])
def test_captured_step_with_feature_tiles_is_bitwise_the_eager_one(gpu, D):
    """32 < D <= 128: the plane image with feature tiles (csrc/glm_planes16d.h) inside a captured step --
    its finalize is a phase of the chained tail (record shape DT x 2), the guide draw stays its own
    launch (only the D <= 32 kernel draws its own weights)."""
    import pyro_amd as pyro
    from pyro_amd import examples
    from pyro_amd.infer import SVI, Trace_ELBO
    from pyro_amd.infer.autoguide import AutoNormal

    out = []
    for graph in (False, True):
        X, y = examples.synthetic_logreg_data(20000, D, gpu, seed=3)
        pyro.clear_param_store()
        pyro.set_rng_seed(5)
        pyro.enable_validation(False)
        guide = AutoNormal(examples.logreg_model, init_scale=0.1)
        svi = SVI(examples.logreg_model, guide, pyro.optim.Adam({"lr": 0.02}),
                  Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1),
                  hip_graph=graph, graph_warmup=3)
        losses = [svi.step(X, y) for _ in range(9)]
        if graph:
            assert svi.hip_graph and len(svi._graphs) == 1 and svi.chain_stats == [(1, 4)], svi.chain_stats
        out.append((losses, _params(pyro)))
    assert out[0][0] == out[1][0]
    for k in out[0][1]:
        assert torch.equal(out[0][1][k], out[1][1][k]), k
