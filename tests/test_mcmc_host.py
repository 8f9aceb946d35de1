"""Host side of the HMC/NUTS path on CPU: adaptation pieces against the reference's golden
vectors, the recursion->iteration transformation of the NUTS tree (oracle vs oracle), and the
MCMC driver with kernels answered by the numpy oracle (tests/oracle_backend.py)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import integrator as o_int
from oracle import nuts as o_nuts
from oracle import nuts_tree as o_tree
from tests import mcmc_cases as mc

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


def test_adaptation_schedule_matches_reference():
    from pyro_amd.infer.mcmc.adaptation import build_adaptation_schedule
    g = load("adaptation")
    for w in (10, 19, 20, 100, 150, 200, 1000):
        got = np.array([[s.start, s.end] for s in build_adaptation_schedule(w)])
        np.testing.assert_array_equal(got, g["schedule/%d" % w])


def test_dual_averaging_matches_reference_scalar_and_batched():
    from pyro_amd.ops.dual_averaging import DualAveraging
    g = load("adaptation")
    da = DualAveraging(prox_center=float(g["dual/prox_center"]))
    # batched: three chains fed the same statistic must reproduce the scalar scheme per chain
    db = DualAveraging(prox_center=torch.full((3,), float(g["dual/prox_center"]),
                                              dtype=torch.float64))
    for i, gi in enumerate(g["dual/g"]):
        da.step(float(gi))
        db.step(torch.full((3,), float(gi), dtype=torch.float64))
        np.testing.assert_allclose(da.get_state(), g["dual/x"][i], rtol=1e-12)
        np.testing.assert_allclose(db.get_state()[0].numpy(), g["dual/x"][i][0], rtol=1e-12)
        np.testing.assert_allclose(db.get_state()[1].numpy(), g["dual/x"][i][1], rtol=1e-12)


def test_welford_matches_reference_single_and_batched():
    from pyro_amd.ops.welford import WelfordCovariance
    g = load("adaptation")
    w1, wb = WelfordCovariance(diagonal=True), WelfordCovariance(diagonal=True)
    for s in g["welford/samples"]:
        w1.update(torch.tensor(s))
        wb.update(torch.tensor(np.stack([s, 2 * s])))
    np.testing.assert_allclose(w1.get_covariance(True).numpy(), g["welford/cov_reg"], rtol=1e-12)
    np.testing.assert_allclose(w1.get_covariance(False).numpy(), g["welford/cov"], rtol=1e-12)
    np.testing.assert_allclose(wb.get_covariance(False)[0].numpy(), g["welford/cov"], rtol=1e-12)
    np.testing.assert_allclose(wb.get_covariance(False)[1].numpy(), 4 * g["welford/cov"], rtol=1e-12)
    wd = WelfordCovariance(diagonal=False)
    for s in g["welford/samples"]:
        wd.update(torch.tensor(s))
    np.testing.assert_allclose(np.diag(wd.get_covariance(False).numpy()), g["welford/cov"],
                               rtol=1e-10)


@pytest.mark.parametrize("D,multinomial", [(5, True), (30, True), (12, False)])
def test_iterative_tree_equals_recursive_reference_formulation(D, multinomial):
    """oracle/nuts_tree.py (iterative state machine = the HIP kernel's formulation) against
    oracle/nuts.py (recursive, pinned against unmodified pyro in test_oracle_vs_golden.py)."""
    rng = np.random.default_rng(0)
    Lam = mc.make_precision(D, 1)
    pg = o_int.gaussian_potential(Lam)
    C = 5
    z = rng.standard_normal((C, D)) * 0.5
    im = rng.uniform(0.5, 1.5, (C, D))
    step = rng.uniform(0.1, 0.4, C)
    pe = np.array([pg(z[c])[0] for c in range(C)])
    g = np.array([pg(z[c])[1] for c in range(C)])
    z2, pe2, g2 = z.copy(), pe.copy(), g.copy()
    tree = o_tree.NutsTreeOracle(z2, pe2, g2, im, step, 8, multinomial, seed=7, chain_offset=3)
    for t in range(3):
        refs = [o_nuts.nuts_transition(z[c], pe[c], g[c], pg, im[c], step[c],
                                       o_nuts.KeyedDraws(7, 3 + c, t, np.float64), 8, multinomial)
                for c in range(C)]
        tree.begin(t)
        while tree.n_active():
            pq = np.array([pg(tree.zq[c])[0] for c in range(C)])
            gq = np.array([pg(tree.zq[c])[1] for c in range(C)])
            tree.advance(pq, gq)
        for c in range(C):
            r = refs[c]
            assert tuple(tree.ints[:, c]) == (r["n_leapfrog"], r["depth"], int(r["diverging"]),
                                              int(r["accepted"]))
            np.testing.assert_allclose(z2[c], r["z"], rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(tree.accept_prob[c], r["accept_prob"], rtol=1e-10)
            z[c], pe[c], g[c] = r["z"], r["pe"], r["grad"]


@pytest.fixture
def _cpu_backend(oracle_backend):
    torch.set_default_dtype(torch.float64)
    yield
    torch.set_default_dtype(torch.float32)


def test_velocity_verlet_matches_reference(_cpu_backend):
    mc.run_integrator_golden(torch.device("cpu"), 1e-10)


@pytest.mark.parametrize("kernel_path", [False, True], ids=["callable", "diag_mass_kernels"])
@pytest.mark.parametrize("system", ["harmonic", "circular", "quartic"])
def test_reference_integrator_kats(_cpu_backend, system, kernel_path):
    from tests import integrator_kat_cases as ik
    ik.run_system(system, torch.device("cpu"), kernel_path)


@pytest.mark.parametrize("kind,multinomial,fused", [("gaussian", True, True),
                                                     ("gaussian", True, False),
                                                     ("logcosh", False, False)])
def test_mcmc_driver_chain_for_chain(_cpu_backend, kind, multinomial, fused):
    mc.run_nuts_chains_vs_oracle(torch.device("cpu"), 6, 3, kind, multinomial, 3, fused=fused)


def test_mcmc_model_potential_and_diagnostics(_cpu_backend):
    """NUTS(model) on a tiny conjugate model through initialize_model (chain-batched potential
    under the _num_chains plate): potential equals the hand-written one, sampler runs with
    adaptation, diagnostics have the reference's keys."""
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    from pyro_amd.infer.mcmc import MCMC, NUTS, initialize_model

    data = torch.tensor([0.3, -0.2, 1.1, 0.7])

    def model(data):
        mu = pyro.sample("mu", dist.Normal(torch.zeros(2), 1.0).to_event(1))
        s = pyro.sample("s", dist.LogNormal(torch.tensor(0.0), 0.5))
        with pyro.plate("d", 4):
            pyro.sample("x", dist.Normal(mu.sum(-1), s), obs=data)

    pyro.set_rng_seed(0)
    init, pot, transforms, _ = initialize_model(model, (data,), num_chains=3)
    assert init["mu"].shape == (3, 2) and init["s"].shape == (3,)
    pe = pot(init)
    assert pe.shape == (3,)
    for c in range(3):
        mu, ls = init["mu"][c], init["s"][c]
        s = ls.exp()
        lp = torch.distributions.Normal(0.0, 1.0).log_prob(mu).sum() \
            + torch.distributions.LogNormal(0.0, 0.5).log_prob(s) \
            + torch.distributions.Normal(mu.sum(), s).log_prob(data).sum() + ls  # + log|dJ|
        np.testing.assert_allclose(pe[c].item(), -lp.item(), rtol=1e-10)
    mcmc = MCMC(NUTS(model, max_tree_depth=4), num_samples=8, warmup_steps=25, num_chains=3)
    mcmc.run(data)
    s = mcmc.get_samples(group_by_chain=True)
    assert s["mu"].shape == (3, 8, 2) and s["s"].shape == (3, 8) and bool((s["s"] > 0).all())
    d = mcmc.diagnostics()
    assert set(d["mu"]) == {"n_eff", "r_hat"} and "divergences" in d and "acceptance rate" in d
    assert mcmc.get_samples()["mu"].shape == (24, 2)


def test_welford_dense_matches_reference():
    from pyro_amd.ops.welford import WelfordCovariance
    g = load("adaptation")
    w1, wb = WelfordCovariance(diagonal=False), WelfordCovariance(diagonal=False)
    for s in g["welford_dense/samples"]:
        w1.update(torch.tensor(s))
        wb.update(torch.tensor(np.stack([s, -s])))
    np.testing.assert_allclose(w1.get_covariance(True).numpy(), g["welford_dense/cov_reg"], rtol=1e-11)
    np.testing.assert_allclose(w1.get_covariance(False).numpy(), g["welford_dense/cov"], rtol=1e-11)
    for c in range(2):
        np.testing.assert_allclose(wb.get_covariance(True)[c].numpy(), g["welford_dense/cov_reg"],
                                   rtol=1e-11)


def test_dense_mass_matrix_products_match_reference(_cpu_backend):
    mc.run_dense_mass_products_vs_reference(torch.device("cpu"))


@pytest.mark.parametrize("kind,multinomial", [("gaussian", True), ("logcosh", False)])
def test_nuts_dense_mass_chain_for_chain(_cpu_backend, kind, multinomial):
    mc.run_nuts_dense_mass_vs_oracle(torch.device("cpu"), 6, 3, kind, multinomial, 3)


def test_dense_mass_adaptation_runs_on_host(_cpu_backend):
    """Host logic of the dense warm-up (window ends re-whiten the state, step-size search in the
    new coordinates): small run through the oracle kernels, structural checks only."""
    out = mc.run_dense_mass_adaptation(torch.device("cpu"), torch.float64, C=2, D=3, warmup=40,
                                       S=5, check=False)
    for V, x in out.values():
        assert V.shape == (2, 3, 3) and np.isfinite(V).all() and np.isfinite(x).all()
        assert np.abs(V - np.eye(3)).max() > 1e-3          # adaptation replaced the identity


def test_structured_mass_host_logic(_cpu_backend):
    """Block structure and API shapes through the host logic (short warm-up; the statistical check
    with the reference's 1000 warm-up steps runs on the device)."""
    try:
        mc.run_structured_mass(torch.device("cpu"), dtype=torch.float64, warmup=60, C=2)
    except AssertionError as e:
        if "not close" not in str(e):       # the covariance tolerance needs the long warm-up
            raise


def test_persistent_launch_path_equals_per_transition_path(_cpu_backend):
    mc.run_persistent_equals_stepwise(torch.device("cpu"), torch.float64, 1e-10, C=3, D=5,
                                      warmup=30, S=4)


@pytest.mark.parametrize("multinomial", [True, False])
def test_asynchronous_chains_equal_lock_step_chains(_cpu_backend, multinomial):
    mc.run_async_equals_lockstep(torch.device("cpu"), torch.float64, 1e-10, C=3, D=5, warmup=30, S=4,
                                 multinomial=multinomial)


def test_compacted_rounds_leave_the_chains_unchanged(_cpu_backend):
    """Late in a span the potential is evaluated at the cursors of the chains still building trees only
    (round sizes C/2, C/4): the same chains as the lock-step schedule."""
    # (to rounding: the batched potential of another batch size rounds its matrix product differently)
    mc.run_async_equals_lockstep(torch.device("cpu"), torch.float64, 1e-7, C=8, D=5, warmup=30, S=6,
                                 min_slots=2)


def test_asynchronous_chains_without_adaptation_are_the_lock_step_chains_exactly(_cpu_backend):
    mc.run_async_equals_lockstep(torch.device("cpu"), torch.float64, 0.0, C=3, D=5, warmup=3, S=8, adapt=False)


def test_discrete_latents_are_summed_out_of_the_potential(_cpu_backend):
    mc.run_enum_potential_vs_reference(torch.device("cpu"))


def test_bernoulli_latent_model_kat(_cpu_backend):
    mc.run_bernoulli_latent_kat(torch.device("cpu"), dtype=torch.float64, C=2)


def test_constrained_support_potentials_match_reference(_cpu_backend):
    mc.run_constrained_potentials_vs_reference(torch.device("cpu"))


def test_sequential_consistent(_cpu_backend):
    mc.run_sequential_consistent(torch.device("cpu"))



def test_initialize_model_redraws_non_finite_starting_points(_cpu_backend):
    """Half of the uniform(-2, 2) initial draws put sqrt(x) at NaN: those chains are drawn again
    until potential and gradient are finite; a model that is never finite raises the reference's
    error (pyro/infer/mcmc/util.py:430-470)."""
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    from pyro_amd.infer.mcmc.util import initialize_model

    def model(bad=False):
        x = pyro.sample("x", dist.Normal(0.0, 3.0))
        loc = torch.sqrt(x) if not bad else torch.log(-x * x - 1.0)
        pyro.sample("obs", dist.Normal(loc, 1.0), obs=torch.tensor(0.5))

    from pyro_amd.primitives import validation_enabled
    pyro.set_rng_seed(0)
    with validation_enabled(False):       # Normal(loc=nan) is what the bad starts produce
        init, potential, _, _ = initialize_model(model, num_chains=16)
        assert bool((init["x"] > 0).all())
        pe = potential({k: v.clone() for k, v in init.items()})
        assert bool(torch.isfinite(pe).all())
        with pytest.raises(ValueError, match="cannot find valid initial params"):
            initialize_model(model, model_args=(True,), num_chains=4)


@pytest.mark.parametrize("tag", ["adapted", "not_pd"])
def test_arrowhead_mass_matrix_matches_reference(monkeypatch, tag):
    """ArrowheadMassMatrix (adaptation.py:395-580): the mass matrix adapted from the reference's
    gradient samples (Welford on gradients, arrowhead mask, regularisation), its inverse -- also when
    the head-tail block has to be halved for positive definiteness -- and kinetic_grad against the
    reference's own objects (tests/golden/arrowhead.npz).  scale / unscale use another square root
    of the same matrix (Cholesky of M^-1 instead of the upper-triangular arrowhead root): they are
    checked as inverses of each other and through the quadratic form they define."""
    from tests import oracle_backend
    from pyro_amd.infer.mcmc.adaptation import ArrowheadDenseMassMatrix
    oracle_backend.install(monkeypatch)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "arrowhead.npz"))
    D, h, C = 6, 4, 2
    dev = torch.device("cpu")
    mm = ArrowheadDenseMassMatrix(C, D, torch.float64, dev, list(range(h)), list(range(h, D)))
    if tag == "adapted":
        for row in g["grads"]:
            mm.update(torch.tensor(row).expand(C, D))
        mm.end_adaptation()
    else:
        M = torch.eye(D, dtype=torch.float64)
        M[:h, h:] = 0.9
        M[h:, :h] = 0.9
        mm.mass_matrix_dense = M.expand(C, D, D)
    top, bottom = mm.mass_matrix
    for c in range(C):
        np.testing.assert_allclose(top[c].numpy(), g[tag + "/top"], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(bottom[c].numpy(), g[tag + "/bottom_diag"], rtol=1e-10)
        np.testing.assert_allclose(mm.inverse_mass_matrix[c].numpy(), g[tag + "/inverse_mass"],
                                   rtol=1e-8, atol=1e-10)
    r = torch.tensor(g[tag + "/r"]).expand(C, D).contiguous()
    np.testing.assert_allclose(mm.kinetic_grad(r)[0].numpy(), g[tag + "/kinetic_grad"], rtol=1e-8)
    u = mm.unscale(r)
    np.testing.assert_allclose((u[0] ** 2).sum().item(), (g[tag + "/unscale"] ** 2).sum(), rtol=1e-8)
    np.testing.assert_allclose(mm.scale(u)[1].numpy(), r[1].numpy(), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(mm.color(mm.whiten(r))[0].numpy(), r[0].numpy(), rtol=1e-8, atol=1e-10)


# ---- the potential of models with discrete latents, and its pickling (tests/infer/mcmc/test_valid_models.py) ----
def _enum_model(data):
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    p = pyro.sample("p", dist.Uniform(0.0, 1.0))
    y = pyro.sample("y", dist.Bernoulli(p))
    q = 0.5 + 0.25 * y
    with pyro.plate("intermediate", 1, dim=-2):
        v = pyro.sample("v", dist.Bernoulli(q))
        with pyro.plate("data", len(data), dim=-1):
            r = 0.4 + 0.1 * v
            z = pyro.sample("z", dist.Bernoulli(r))
            pyro.sample("obs", dist.Normal(2 * z - 1, 1.0), obs=data)


@pytest.mark.parametrize("data,expected_log_prob", [
    (torch.tensor([1.0]), torch.tensor(-1.3434)),
    (torch.tensor([0.0]), torch.tensor(-1.4189)),
    (torch.tensor([1.0, 0.0]), torch.tensor(-3.1767)),
])
def test_trace_einsum_evaluator_sums_out_enumerated_sites(oracle_backend, data, expected_log_prob):
    """test_enum_log_prob_nested_plates-style known answers of the reference: p conditioned, y / v / z
    enumerated in parallel, the log joint with all three summed out."""
    from pyro_amd import poutine
    from pyro_amd.infer import config_enumerate
    from pyro_amd.infer.mcmc.util import TraceEinsumEvaluator, TraceTreeEvaluator
    model = poutine.enum(config_enumerate(poutine.condition(_enum_model, data={"p": torch.tensor(0.4)})),
                         first_available_dim=-3)
    tr = poutine.trace(model).get_trace(data)
    for Eval in (TraceEinsumEvaluator, TraceTreeEvaluator):
        lp = Eval(tr, True, 2).log_prob(tr)
        # brute force over the 2 x 2 x 2^n assignments
        import itertools
        import math
        total = -math.inf
        n = len(data)
        for y, v in itertools.product((0.0, 1.0), repeat=2):
            q = 0.5 + 0.25 * y
            r = 0.4 + 0.1 * v
            lp_yv = math.log(0.4 if y else 0.6) + math.log(q if v else 1 - q)
            for zs in itertools.product((0.0, 1.0), repeat=n):
                t = lp_yv
                for z, x in zip(zs, data.tolist()):
                    t += math.log(r if z else 1 - r) - 0.5 * (x - (2 * z - 1)) ** 2 - 0.5 * math.log(2 * math.pi)
                total = max(total, t) + math.log1p(math.exp(-abs(total - t))) if total > -math.inf else t
        assert abs(float(lp) - total) < 1e-5, (float(lp), total)
    with pytest.raises(ValueError, match="Finite value required for `max_plate_nesting`"):
        TraceEinsumEvaluator(tr, True, None)


def _beta_bernoulli_model(data):
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    p = pyro.sample("p_latent", dist.Beta(torch.tensor([1.1, 1.1]), torch.tensor([1.1, 1.1])))
    with pyro.plate("data", data.shape[0], dim=-2):
        pyro.sample("obs", dist.Bernoulli(p), obs=data)


def test_potential_fn_survives_torch_save(oracle_backend):
    import io
    from pyro_amd.infer.mcmc.util import initialize_model
    torch.manual_seed(0)
    data = (torch.rand(50, 2) < torch.tensor([0.8, 0.2])).float()
    _, potential_fn, _, _ = initialize_model(_beta_bernoulli_model, (data,), jit_compile=True,
                                             skip_jit_warnings=True)
    buffer = io.BytesIO()
    torch.save(potential_fn, buffer)
    buffer.seek(0)
    again = torch.load(buffer, weights_only=False)
    z = {"p_latent": torch.tensor([0.2, 0.6])}
    assert torch.allclose(again(z), potential_fn(z))


def test_initialize_model_checks_shapes_when_discrete_sites_are_summed_out(oracle_backend):
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    from pyro_amd.infer.mcmc.util import initialize_model

    def gmm():
        data = torch.tensor([0.0, 0.0, 3.0, 3.0, 3.0, 5.0, 5.0])
        mix = pyro.sample("phi", dist.Dirichlet(torch.ones(3)))
        means = pyro.sample("cluster_means", dist.Normal(torch.arange(3.0), 1.0))   # stray batch dim
        with pyro.plate("data", 7):
            a = pyro.sample("assignments", dist.Categorical(mix))
            pyro.sample("obs", dist.Normal(means[a], 1.0), obs=data)

    pyro.enable_validation(True)
    try:
        with pytest.raises(ValueError, match="invalid log_prob shape"):
            initialize_model(gmm)
    finally:
        pyro.enable_validation(False)


def test_clear_cache_restarts_the_kernel_at_the_given_point(oracle_backend):
    """tests/infer/mcmc/test_hmc.py:309-337 drives a kernel by hand: kernel(params) is one transition from
    the cached state, and after clear_cache() from ``params``."""
    import pyro_amd as pyro
    from pyro_amd.infer import HMC
    pyro.set_rng_seed(0)

    def potential(params):
        return 0.5 * torch.sum((params["z"] - 5.0) ** 2)

    kernel = HMC(potential_fn=potential, step_size=1e-3, num_steps=2, adapt_step_size=False,
                 adapt_mass_matrix=False)
    kernel.initial_params = {"z": torch.tensor(0.0)}
    kernel.setup(0)
    a = kernel({"z": torch.tensor(0.0)})
    b = kernel({"z": torch.tensor(100.0)})           # ignored: continues from the cached state
    assert abs(float(b["z"]) - float(a["z"])) < 0.1
    kernel.clear_cache()
    c = kernel({"z": torch.tensor(100.0)})           # restarts there
    assert abs(float(c["z"]) - 100.0) < 1.0
    d = kernel(c)
    assert abs(float(d["z"]) - float(c["z"])) < 1.0


def test_random_walk_kernel_on_a_conjugate_model(oracle_backend):
    """tests/infer/mcmc/test_rwkernel.py: Beta-Bernoulli posterior mean and variance."""
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    from pyro_amd.infer import MCMC, RandomWalkKernel
    pyro.set_rng_seed(0)
    torch.manual_seed(0)
    alpha = beta = torch.tensor([1.1, 2.2])

    def model(data):
        p = pyro.sample("p_latent", dist.Beta(alpha, beta))
        with pyro.plate("data", data.shape[0], dim=-2):
            pyro.sample("obs", dist.Bernoulli(p), obs=data)

    data = torch.tensor([[1.0, 0.0], [1.0, 0.0], [1.0, 0.0], [1.0, 1.0], [0.0, 0.0]])
    kernel = RandomWalkKernel(model)
    mcmc = MCMC(kernel, num_samples=3000, warmup_steps=500)
    mcmc.run(data)
    s = mcmc.get_samples()["p_latent"]
    a = alpha + data.sum(0)
    b = beta + 5 - data.sum(0)
    mean = a / (a + b)
    var = mean.pow(2) * b / (a * (1 + a + b))
    assert torch.allclose(s.mean(0), mean, atol=0.04) and torch.allclose(s.var(0), var, atol=0.008)
    rate = mcmc.diagnostics()["acceptance rate"]["chain 0"]
    assert 0.05 < rate < 0.95
    for bad in (dict(init_step_size=0), dict(target_accept_prob=1.0)):
        with pytest.raises(ValueError):
            RandomWalkKernel(model, **bad)
