"""TraceEnum_ELBO known-answer tests of the reference (tests/infer/test_enum.py:1795-2632:
test_elbo_enumerate_1..3, _plate_1..3, _plates_1..2), restated compactly against the drop-in API.
Each case builds an "auto" model that relies on plated parallel enumeration and a "hand" model
in which the same quantity is written out (closed-form marginal, or sequential plates), and
requires equal loss and equal gradients w.r.t. every parameter -- _check_loss_and_grads of the
reference.  The guide enumerates its discrete site (config_enumerate), so the DiCE expectation
over guide enumeration is part of every case that has a guide."""
import torch
from torch.autograd import grad

import pyro_amd as pyro
import pyro_amd.distributions as dist
import pyro_amd.poutine as poutine
from pyro_amd.distributions import constraints
from pyro_amd.infer import TraceEnum_ELBO, config_enumerate


def _check_loss_and_grads(expected_loss, actual_loss):
    torch.testing.assert_close(actual_loss, expected_loss, rtol=1e-5, atol=1e-6)
    names = sorted(pyro.get_param_store().keys())
    params = [pyro.param(name).unconstrained() for name in names]
    actual = grad(actual_loss, params, allow_unused=True, retain_graph=True)
    expected = grad(expected_loss, params, allow_unused=True, retain_graph=True)
    for name, a, e in zip(names, actual, expected):
        if a is None or e is None:
            continue
        torch.testing.assert_close(a, e, rtol=1e-4, atol=1e-6, msg=lambda m, n=name: n + ": " + m)


def _xyz_params(device, z_depends_on_y=True):
    def t(v):
        return torch.tensor(v, device=device)
    pyro.clear_param_store()
    pyro.param("guide_probs_x", t([0.1, 0.9]), constraint=constraints.simplex)
    pyro.param("model_probs_x", t([0.4, 0.6]), constraint=constraints.simplex)
    pyro.param("model_probs_y", t([[0.75, 0.25], [0.55, 0.45]]), constraint=constraints.simplex)
    pyro.param("model_probs_z", t([[0.3, 0.7], [0.2, 0.8]]) if z_depends_on_y else t([0.3, 0.7]),
               constraint=constraints.simplex)


PAR = {"enumerate": "parallel"}


def run_enumerate_chain(device, variant, scale):
    """x -> y -> z without plates (test_elbo_enumerate_1..3): y enumerated in the model and summed
    out; the hand model carries the closed-form marginal probs_y @ probs_z."""
    _xyz_params(device, z_depends_on_y=variant != 1)
    zero = torch.tensor(0, device=device)

    def auto_model():
        px, py, pz = (pyro.param("model_probs_" + k) for k in "xyz")
        x = pyro.sample("x", dist.Categorical(px))
        with poutine.scale(scale=scale if variant == 3 else 1.0):
            y = pyro.sample("y", dist.Categorical(py[x]), infer=PAR)
            pyro.sample("z", dist.Categorical(pz[y] if variant != 1 else pz), obs=zero)

    def hand_model():
        px, py, pz = (pyro.param("model_probs_" + k) for k in "xyz")
        x = pyro.sample("x", dist.Categorical(px))
        with poutine.scale(scale=scale if variant == 3 else 1.0):
            pyro.sample("z", dist.Categorical(py.mm(pz)[x] if variant != 1 else pz), obs=zero)

    def guide():
        pyro.sample("x", dist.Categorical(pyro.param("guide_probs_x")))

    guide = config_enumerate(guide)
    if variant != 3:           # the whole program scaled (models and guide alike)
        auto_model, hand_model, guide = (poutine.scale(f, scale=scale)
                                         for f in (auto_model, hand_model, guide))
    elbo = TraceEnum_ELBO(max_plate_nesting=0, strict_enumeration_warning=False)
    _check_loss_and_grads(elbo.differentiable_loss(hand_model, guide),
                          elbo.differentiable_loss(auto_model, guide))


def run_enumerate_plate(device, variant, num_samples, num_masked, scale):
    """x -> y -> z with a data plate around z only (1), around y and z (2), around everything (3)
    (test_elbo_enumerate_plate_1..3), optionally masked; hand model: a sequential plate over the
    unmasked indices."""
    _xyz_params(device)
    g = torch.Generator().manual_seed(num_samples)
    data = torch.multinomial(torch.tensor([0.3, 0.7]), num_samples, replacement=True,
                             generator=g).to(device)
    masked = num_masked != num_samples
    mask = torch.arange(num_samples, device=device) < num_masked

    def maybe_mask():
        return poutine.mask(mask=mask) if masked else poutine.scale(scale=1.0)

    def auto_model(data):
        px, py, pz = (pyro.param("model_probs_" + k) for k in "xyz")
        if variant == 3:
            with pyro.plate("data", len(data)), maybe_mask():
                x = pyro.sample("x", dist.Categorical(px))
                y = pyro.sample("y", dist.Categorical(py[x]), infer=PAR)
                pyro.sample("z", dist.Categorical(pz[y]), obs=data)
            return
        x = pyro.sample("x", dist.Categorical(px))
        with poutine.scale(scale=scale):
            if variant == 1:
                y = pyro.sample("y", dist.Categorical(py[x]), infer=PAR)
                with pyro.plate("data", len(data)), maybe_mask():
                    pyro.sample("z", dist.Categorical(pz[y]), obs=data)
            else:
                with pyro.plate("data", len(data)), maybe_mask():
                    y = pyro.sample("y", dist.Categorical(py[x]), infer=PAR)
                    pyro.sample("z", dist.Categorical(pz[y]), obs=data)

    def hand_model(data):
        px, py, pz = (pyro.param("model_probs_" + k) for k in "xyz")
        if variant == 3:
            for i in pyro.plate("data", num_masked):
                x = pyro.sample("x_{}".format(i), dist.Categorical(px))
                y = pyro.sample("y_{}".format(i), dist.Categorical(py[x]), infer=PAR)
                pyro.sample("z_{}".format(i), dist.Categorical(pz[y]), obs=data[i])
            return
        x = pyro.sample("x", dist.Categorical(px))
        with poutine.scale(scale=scale):
            if variant == 1:
                y = pyro.sample("y", dist.Categorical(py[x]), infer=PAR)
            for i in pyro.plate("data", num_masked):
                if variant == 2:
                    y = pyro.sample("y_{}".format(i), dist.Categorical(py[x]), infer=PAR)
                pyro.sample("z_{}".format(i), dist.Categorical(pz[y]), obs=data[i])

    def auto_guide(data):
        pq = pyro.param("guide_probs_x")
        if variant == 3:
            with pyro.plate("data", len(data)), maybe_mask():
                pyro.sample("x", dist.Categorical(pq))
        else:
            pyro.sample("x", dist.Categorical(pq))

    def hand_guide(data):
        pq = pyro.param("guide_probs_x")
        if variant == 3:
            for i in pyro.plate("data", num_masked):
                pyro.sample("x_{}".format(i), dist.Categorical(pq))
        else:
            pyro.sample("x", dist.Categorical(pq))

    auto_guide, hand_guide = config_enumerate(auto_guide), config_enumerate(hand_guide)
    if variant == 3:
        auto_model, hand_model, auto_guide, hand_guide = (
            poutine.scale(f, scale=scale) for f in (auto_model, hand_model, auto_guide, hand_guide))
    auto = TraceEnum_ELBO(max_plate_nesting=1, strict_enumeration_warning=False)
    hand = TraceEnum_ELBO(max_plate_nesting=1 if variant != 1 else 0,
                          strict_enumeration_warning=False)
    _check_loss_and_grads(hand.differentiable_loss(hand_model, hand_guide, data),
                          auto.differentiable_loss(auto_model, auto_guide, data))


def run_enumerate_plates(device, variant, scale):
    """Two plates: unrelated (1: a->b in M, c->d in N) or sharing an enumerated parent
    (2: b <- a -> c), everything enumerated in the model, empty guide
    (test_elbo_enumerate_plates_1..2)."""
    def t(v):
        return torch.tensor(v, device=device)
    pyro.clear_param_store()
    pyro.param("probs_a", t([0.45, 0.55]), constraint=constraints.simplex)
    pyro.param("probs_b", t([[0.6, 0.4], [0.4, 0.6]]), constraint=constraints.simplex)
    if variant == 1:
        pyro.param("probs_c", t([0.75, 0.25]), constraint=constraints.simplex)
        pyro.param("probs_d", t([[0.4, 0.6], [0.3, 0.7]]), constraint=constraints.simplex)
    else:
        pyro.param("probs_c", t([[0.75, 0.25], [0.55, 0.45]]), constraint=constraints.simplex)
    b_data, cd_data = t([0, 1]), t([0, 0, 1])

    def auto_model():
        pa, pb, pc = (pyro.param("probs_" + k) for k in "abc")
        if variant == 1:
            pd = pyro.param("probs_d")
            with pyro.plate("a_axis", 2):
                a = pyro.sample("a", dist.Categorical(pa))
                pyro.sample("b", dist.Categorical(pb[a]), obs=b_data)
            with pyro.plate("c_axis", 3):
                c = pyro.sample("c", dist.Categorical(pc))
                pyro.sample("d", dist.Categorical(pd[c]), obs=cd_data)
        else:
            a = pyro.sample("a", dist.Categorical(pa))
            with pyro.plate("b_axis", 2):
                pyro.sample("b", dist.Categorical(pb[a]), obs=b_data)
            with pyro.plate("c_axis", 3):
                pyro.sample("c", dist.Categorical(pc[a]), obs=cd_data)

    def hand_model():
        pa, pb, pc = (pyro.param("probs_" + k) for k in "abc")
        if variant == 1:
            pd = pyro.param("probs_d")
            for i in pyro.plate("a_axis", 2):
                a = pyro.sample("a_{}".format(i), dist.Categorical(pa))
                pyro.sample("b_{}".format(i), dist.Categorical(pb[a]), obs=b_data[i])
            for j in pyro.plate("c_axis", 3):
                c = pyro.sample("c_{}".format(j), dist.Categorical(pc))
                pyro.sample("d_{}".format(j), dist.Categorical(pd[c]), obs=cd_data[j])
        else:
            a = pyro.sample("a", dist.Categorical(pa))
            for i in pyro.plate("b_axis", 2):
                pyro.sample("b_{}".format(i), dist.Categorical(pb[a]), obs=b_data[i])
            for j in pyro.plate("c_axis", 3):
                pyro.sample("c_{}".format(j), dist.Categorical(pc[a]), obs=cd_data[j])

    def guide():
        pass

    auto_model = config_enumerate(poutine.scale(auto_model, scale=scale))
    hand_model = config_enumerate(poutine.scale(hand_model, scale=scale))
    auto_loss = TraceEnum_ELBO(max_plate_nesting=1).differentiable_loss(auto_model, guide)
    hand_loss = TraceEnum_ELBO(max_plate_nesting=0).differentiable_loss(hand_model, guide)
    _check_loss_and_grads(hand_loss, auto_loss)


def run_guide_enumeration_closed_form(device):
    """Guide-side enumeration gives the EXACT expectation: for a discrete latent with an observed
    child the ELBO equals sum_x q(x) [log p(x) + log p(obs | x) - log q(x)] (the KL-type closed
    forms of tests/infer/test_enum.py:341-395), with zero variance across seeds."""
    def t(v):
        return torch.tensor(v, device=device)
    pyro.clear_param_store()
    q = pyro.param("q", t([0.2, 0.5, 0.3]), constraint=constraints.simplex)
    p = t([0.5, 0.25, 0.25])
    lik = t([[0.9, 0.1], [0.4, 0.6], [0.2, 0.8]])

    def model():
        x = pyro.sample("x", dist.Categorical(p))
        with pyro.plate("d", 3):
            pyro.sample("obs", dist.Categorical(lik[x]), obs=t([1, 0, 1]))

    @config_enumerate
    def guide():
        pyro.sample("x", dist.Categorical(pyro.param("q")))

    elbo = TraceEnum_ELBO(max_plate_nesting=1)
    losses = []
    for seed in (0, 1):
        pyro.set_rng_seed(seed)
        losses.append(elbo.differentiable_loss(model, guide))
    assert torch.equal(losses[0].detach(), losses[1].detach())
    qq = pyro.param("q")
    ll = lik[:, 1].log() * 2 + lik[:, 0].log()
    expected = -(qq * (p.log() + ll - qq.log())).sum()
    torch.testing.assert_close(losses[0], expected, rtol=1e-5, atol=1e-6)
    u = pyro.param("q").unconstrained()
    ga, = grad(losses[0], [u], retain_graph=True)
    ge, = grad(expected, [u])
    torch.testing.assert_close(ga, ge, rtol=1e-4, atol=1e-6)


def run_sequential_equals_parallel(device):
    """Guide-side enumeration strategy does not change the estimate: "sequential" (one trace per
    assignment, pyro/infer/enum.py:88-135) and "parallel" (one tensor dim per site) give the same
    loss and gradients -- two dependent guide sites, one of them inside a plate, plus a
    model-enumerated site (the reference parametrises its plate tests over enumerate1 / enumerate2
    in {sequential, parallel}: tests/infer/test_enum.py:2345-2405)."""
    def t(v):
        return torch.tensor(v, device=device)
    data = t([0, 1, 1])

    def model():
        pa = pyro.param("model_a", t([0.3, 0.7]), constraint=constraints.simplex)
        pb = pyro.param("model_b", t([[0.6, 0.4], [0.1, 0.9]]), constraint=constraints.simplex)
        pc = pyro.param("model_c", t([[0.5, 0.5], [0.2, 0.8]]), constraint=constraints.simplex)
        pz = pyro.param("model_z", t([[0.9, 0.1], [0.3, 0.7]]), constraint=constraints.simplex)
        a = pyro.sample("a", dist.Categorical(pa))
        with pyro.plate("d", 3):
            b = pyro.sample("b", dist.Categorical(pb[a]))
            c = pyro.sample("c", dist.Categorical(pc[b]), infer=PAR)
            pyro.sample("z", dist.Categorical(pz[c]), obs=data)

    def make_guide(how_a, how_b):
        def guide():
            qa = pyro.param("guide_a", t([0.4, 0.6]), constraint=constraints.simplex)
            qb = pyro.param("guide_b", t([[0.7, 0.3], [0.45, 0.55]]), constraint=constraints.simplex)
            a = pyro.sample("a", dist.Categorical(qa), infer={"enumerate": how_a})
            with pyro.plate("d", 3):
                pyro.sample("b", dist.Categorical(qb[a]), infer={"enumerate": how_b})
        return guide

    pyro.clear_param_store()
    elbo = TraceEnum_ELBO(max_plate_nesting=1, strict_enumeration_warning=False)
    ref = elbo.differentiable_loss(model, make_guide("parallel", "parallel"))
    for how_a, how_b in (("sequential", "parallel"), ("sequential", "sequential")):
        if how_b == "sequential":
            # a sequential site inside a vectorised plate enumerates ONE value for the whole plate
            # slice per trace, which is not the per-element expectation: the reference restricts
            # this combination the same way (sequential sites must not be in vectorised plates)
            continue
        loss = elbo.differentiable_loss(model, make_guide(how_a, how_b))
        _check_loss_and_grads(ref, loss)


def run_markov_history(device, history, T=5, K=2):
    """pyro.markov(history=h): x_t depends on the previous h states; the loss of the enumerated
    model equals the brute-force sum over all K^T joint assignments (the reference's
    tests/infer/test_enum.py markov tests compare against hand-unrolled models the same way), and
    no more than h + 1 enumeration dims are ever in use."""
    import itertools
    torch.manual_seed(history)
    init = torch.softmax(torch.randn(K), -1).to(device)
    # transition table indexed by the previous `history` states
    trans = torch.softmax(torch.randn((K,) * history + (K,)), -1).to(device)
    emit = torch.softmax(torch.randn(K, 3), -1).to(device)
    data = torch.randint(0, 3, (T,)).to(device)

    def model():
        xs = []
        for t in pyro.markov(range(T), history=history):
            if t < history:
                probs = init
            else:
                probs = trans[tuple(xs[t - history:t])]
            x = pyro.sample("x_{}".format(t), dist.Categorical(probs), infer=PAR)
            pyro.sample("y_{}".format(t), dist.Categorical(emit[x]), obs=data[t])
            xs.append(x)

    pyro.clear_param_store()
    elbo = TraceEnum_ELBO(max_plate_nesting=0)
    loss = elbo.differentiable_loss(model, lambda: None)
    total = torch.zeros((), dtype=init.dtype, device=device)
    for assign in itertools.product(range(K), repeat=T):
        p = torch.ones((), dtype=init.dtype, device=device)
        for t, x in enumerate(assign):
            p = p * (init[x] if t < history else trans[tuple(assign[t - history:t]) + (x,)])
            p = p * emit[x, data[t]]
        total = total + p
    torch.testing.assert_close(loss, -total.log(), rtol=1e-5, atol=1e-6)
    tr = poutine.trace(poutine.enum(model, first_available_dim=-1)).get_trace()
    dims = {s["infer"]["_enumerate_dim"] for s in tr.nodes.values()
            if s["type"] == "sample" and s["infer"].get("_enumerate_dim") is not None}
    assert len(dims) == history + 1, dims


def run_guide_side_markov(device, T=4, K=2):
    """A Markov chain enumerated in the GUIDE (config_enumerate) under pyro.markov: the ELBO is the
    exact expectation sum_x q(x) [log p(x, y) - log q(x)] over all K^T paths, although the program
    only ever used two enumeration dims."""
    import itertools
    torch.manual_seed(11)
    trans = torch.softmax(torch.randn(K, K), -1).to(device)
    emit = torch.softmax(torch.randn(K, 3), -1).to(device)
    data = torch.randint(0, 3, (T,)).to(device)
    q0 = torch.softmax(torch.randn(K, K), -1).to(device)

    def model():
        x = 0
        for t in pyro.markov(range(T)):
            x = pyro.sample("x_{}".format(t), dist.Categorical(trans[x]))
            pyro.sample("y_{}".format(t), dist.Categorical(emit[x]), obs=data[t])

    @config_enumerate
    def guide():
        q = pyro.param("q", q0, constraint=constraints.simplex)
        x = 0
        for t in pyro.markov(range(T)):
            x = pyro.sample("x_{}".format(t), dist.Categorical(q[x]))

    pyro.clear_param_store()
    loss = TraceEnum_ELBO(max_plate_nesting=0).differentiable_loss(model, guide)
    q = pyro.param("q")
    total = 0.0
    for path in itertools.product(range(K), repeat=T):
        lq, lp, prev = 0.0, 0.0, 0
        for t, x in enumerate(path):
            lq = lq + q[prev, x].log()
            lp = lp + trans[prev, x].log() + emit[x, data[t]].log()
            prev = x
        total = total + lq.exp() * (lp - lq)
    torch.testing.assert_close(loss, -total, rtol=1e-5, atol=1e-6)
    u = q.unconstrained()
    ga, = grad(loss, [u], retain_graph=True)
    ge, = grad(-total, [u])
    torch.testing.assert_close(ga, ge, rtol=1e-4, atol=1e-6)


# ---- closed-form KL known answers (tests/infer/test_enum.py:338-395, :544-622, :625-714) -----------
def run_elbo_bern(device, method, enumerate1, scale):
    """One Bernoulli site enumerated in the guide: loss = scale * KL(q || p) exactly (prec 1e-3),
    through loss / differentiable_loss / loss_and_grads."""
    from torch.distributions import kl_divergence
    pyro.clear_param_store()
    q = pyro.param("q", torch.tensor(0.5, dtype=torch.float64, device=device, requires_grad=True))
    p = torch.tensor(0.25, dtype=torch.float64, device=device)
    kl = kl_divergence(torch.distributions.Bernoulli(q), torch.distributions.Bernoulli(p))

    @poutine.scale(scale=scale)
    def model():
        with pyro.plate("particles", 1):
            pyro.sample("z", dist.Bernoulli(p).expand([1]))

    @config_enumerate(default=enumerate1)
    @poutine.scale(scale=scale)
    def guide():
        qq = pyro.param("q")
        with pyro.plate("particles", 1):
            pyro.sample("z", dist.Bernoulli(qq).expand([1]))

    elbo = TraceEnum_ELBO(strict_enumeration_warning=True)
    if method == "loss":
        assert abs(elbo.loss(model, guide) - kl.item() * scale) < 1e-3
        return
    if method == "differentiable_loss":
        actual = grad(elbo.differentiable_loss(model, guide), [q])[0]
    else:
        elbo.loss_and_grads(model, guide)
        actual = pyro.param("q").unconstrained().grad
    expected = grad(kl, [q])[0] * scale
    assert abs(float(actual) - float(expected)) < 1e-3, (actual, expected)


def run_elbo_berns(device, method, enums):
    """Three Bernoulli sites, each enumerated sequentially or in parallel in the guide: the loss is
    the sum of the three KLs and its gradient w.r.t. the shared q exact (prec 1e-3)."""
    from torch.distributions import kl_divergence
    pyro.clear_param_store()
    q = pyro.param("q", torch.tensor(0.75, dtype=torch.float64, device=device, requires_grad=True))
    ps = [torch.tensor(v, dtype=torch.float64, device=device) for v in (0.1, 0.2, 0.3)]

    def model():
        for i, p in enumerate(ps):
            pyro.sample("x%d" % (i + 1), dist.Bernoulli(p))

    def guide():
        qq = pyro.param("q")
        for i, e in enumerate(enums):
            pyro.sample("x%d" % (i + 1), dist.Bernoulli(qq), infer={"enumerate": e})

    kl = sum(kl_divergence(torch.distributions.Bernoulli(q), torch.distributions.Bernoulli(p))
             for p in ps)
    expected_grad = grad(kl, [q])[0]
    elbo = TraceEnum_ELBO(num_particles=1, vectorize_particles=True, strict_enumeration_warning=True)
    if method == "differentiable_loss":
        loss = elbo.differentiable_loss(model, guide)
        actual_loss, actual_grad = loss.item(), grad(loss, [q])[0]
    else:
        actual_loss = elbo.loss_and_grads(model, guide)
        actual_grad = pyro.param("q").unconstrained().grad
    assert abs(actual_loss - kl.item()) < 1e-3, (actual_loss, kl.item())
    assert abs(float(actual_grad) - float(expected_grad)) < 1e-3, (actual_grad, expected_grad)


def run_elbo_categoricals(device, enums, max_plate_nesting):
    """Three Categorical sites of different support sizes, sequential / parallel in any mix."""
    from torch.distributions import kl_divergence

    def t(v):
        return torch.tensor(v, dtype=torch.float64, device=device)

    pyro.clear_param_store()
    ps = [t([0.6, 0.4]), t([0.3, 0.3, 0.4]), t([0.1, 0.2, 0.3, 0.4])]
    qs = [pyro.param("q%d" % (i + 1), v.requires_grad_(True)) for i, v in enumerate(
        [t([0.4, 0.6]), t([0.4, 0.3, 0.3]), t([0.4, 0.3, 0.2, 0.1])])]

    def model():
        for i, p in enumerate(ps):
            pyro.sample("x%d" % (i + 1), dist.Categorical(p))

    def guide():
        for i, e in enumerate(enums):
            pyro.sample("x%d" % (i + 1), dist.Categorical(pyro.param("q%d" % (i + 1))),
                        infer={"enumerate": e})

    kl = sum(kl_divergence(torch.distributions.Categorical(q), torch.distributions.Categorical(p))
             for q, p in zip(qs, ps))
    expected_grads = grad(kl, qs)
    elbo = TraceEnum_ELBO(max_plate_nesting=max_plate_nesting, strict_enumeration_warning=True)
    actual_loss = elbo.loss_and_grads(model, guide)
    assert abs(actual_loss - kl.item()) < 1e-3, (actual_loss, kl.item())
    for i, e in enumerate(expected_grads):
        a = pyro.param("q%d" % (i + 1)).unconstrained().grad
        assert float((a - e).abs().max()) < 1e-3, (i, a, e)


# ---- shapes under vectorised particles + enumeration (tests/infer/test_valid_models.py:1661-1795) ---
def run_vectorized_num_particles(device, elbo_name):
    from pyro_amd import infer
    Elbo = getattr(infer, elbo_name)
    data = torch.ones(1000, 2, device=device)
    a = torch.tensor(1.1, device=device)

    def model():
        with pyro.plate("components", 2):
            p = pyro.sample("p", dist.Beta(a, a))
            assert p.shape == (10, 1, 2)
            with pyro.plate("data", data.shape[0]):
                pyro.sample("obs", dist.Bernoulli(p), obs=data)

    def guide():
        with pyro.plate("components", 2):
            pyro.sample("p", dist.Beta(a, a))

    pyro.clear_param_store()
    g = config_enumerate(guide) if elbo_name == "TraceEnum_ELBO" else guide
    kw = {"strict_enumeration_warning": False} if elbo_name == "TraceEnum_ELBO" else {}
    loss = Elbo(num_particles=10, vectorize_particles=True, max_plate_nesting=2, **kw).loss(model, g)
    assert loss == loss and abs(loss) != float("inf")


def run_enum_discrete_vectorized_num_particles(device, enumerate_, expand, num_particles):
    a = torch.tensor(1.1, device=device)
    half = torch.tensor(0.5, device=device)
    P = num_particles

    @config_enumerate(default=enumerate_, expand=expand)
    def model():
        x_plate = pyro.plate("x_plate", 10, 5, dim=-1)
        y_plate = pyro.plate("y_plate", 11, 6, dim=-2)
        with x_plate:
            b = pyro.sample("b", dist.Beta(a, a))
        with y_plate:
            c = pyro.sample("c", dist.Bernoulli(half))
        with x_plate, y_plate:
            d = pyro.sample("d", dist.Bernoulli(b))
        lead = (P, 1) if P > 1 else ()
        assert b.shape == lead + (5,) if P > 1 else b.shape == (5,)
        if enumerate_ == "parallel":
            if expand:
                assert c.shape == ((2, P, 6, 1) if P > 1 else (2, 6, 1))
                assert d.shape == ((2, 1, P, 6, 5) if P > 1 else (2, 1, 6, 5))
            else:
                assert c.shape == ((2, 1, 1, 1) if P > 1 else (2, 1, 1))
                assert d.shape == ((2, 1, 1, 1, 1) if P > 1 else (2, 1, 1, 1))
        elif enumerate_ == "sequential":
            if expand:
                assert c.shape == ((P, 6, 1) if P > 1 else (6, 1))
                assert d.shape == ((P, 6, 5) if P > 1 else (6, 5))
            else:
                assert c.shape == ((1, 1, 1) if P > 1 else (1, 1))
                assert d.shape == ((1, 1, 1) if P > 1 else (1, 1))
        else:
            assert c.shape == ((P, 6, 1) if P > 1 else (6, 1))
            assert d.shape == ((P, 6, 5) if P > 1 else (6, 5))

    pyro.clear_param_store()
    pyro.set_rng_seed(0)
    elbo = TraceEnum_ELBO(max_plate_nesting=2, num_particles=P, vectorize_particles=True,
                          strict_enumeration_warning=(enumerate_ == "parallel"))
    loss = elbo.loss(model, model)
    assert loss == loss and abs(loss) != float("inf")


def run_elbo_plate_plate(device, enums):
    """tests/infer/test_enum.py:944-1016 with every guide site enumerated: w outside, x in plate
    "outer", y in plate "inner", z in both.  A sequentially enumerated site splits the run into one
    trace per value, and the costs that are not downstream of it must still count once: the loss is
    (1 + outer + inner + outer*inner) KLs exactly."""
    from torch.distributions import kl_divergence
    pyro.clear_param_store()
    outer_dim, inner_dim = 2, 2
    q = pyro.param("q", torch.tensor(0.75, dtype=torch.float64, device=device, requires_grad=True))
    p = torch.tensor(0.2693204236205713, dtype=torch.float64, device=device)

    def program(d, infers):
        context1 = pyro.plate("outer", outer_dim, dim=-1)
        context2 = pyro.plate("inner", inner_dim, dim=-2)
        pyro.sample("w", d, infer=infers[0])
        with context1:
            pyro.sample("x", d, infer=infers[1])
        with context2:
            pyro.sample("y", d, infer=infers[2])
        with context1, context2:
            pyro.sample("z", d, infer=infers[3])

    def model():
        program(dist.Bernoulli(p), [{}] * 4)

    def guide():
        program(dist.Bernoulli(pyro.param("q")), [{"enumerate": e} for e in enums])

    kl = (1 + outer_dim + inner_dim + outer_dim * inner_dim) * kl_divergence(
        torch.distributions.Bernoulli(q), torch.distributions.Bernoulli(p))
    expected_grad = grad(kl, [q])[0]
    elbo = TraceEnum_ELBO(num_particles=1, vectorize_particles=True, strict_enumeration_warning=True)
    actual_loss = elbo.loss_and_grads(model, guide)
    assert abs(actual_loss - kl.item()) < 1e-6, (actual_loss, kl.item())
    assert abs(float(pyro.param("q").grad) - float(expected_grad)) < 1e-6


def run_local_sampling(device, num_samples=20000, tmc="diagonal", expand=False):
    """``num_samples`` draws on an enumeration dim instead of the support (tests/infer/test_enum.py:
    454-540, 797-870): two dependent Bernoulli sites in a plate, both multiply sampled in the guide.
    The estimate is the exact ELBO up to Monte-Carlo error; the gradient is the score-function one."""
    from torch.distributions import kl_divergence
    pyro.clear_param_store()
    pyro.set_rng_seed(0)
    q = pyro.param("q", torch.tensor(0.75, dtype=torch.float64, device=device, requires_grad=True))
    p = torch.tensor(0.2693204236205713, dtype=torch.float64, device=device)
    infer = {"enumerate": "parallel", "num_samples": num_samples, "tmc": tmc, "expand": expand}

    def model():
        pyro.sample("y", dist.Bernoulli(p))
        with pyro.plate("plate", 3):
            pyro.sample("z", dist.Bernoulli(p))

    def guide():
        qq = pyro.param("q")
        pyro.sample("y", dist.Bernoulli(qq), infer=infer)
        with pyro.plate("plate", 3):
            pyro.sample("z", dist.Bernoulli(qq), infer=infer)

    kl = 4 * kl_divergence(torch.distributions.Bernoulli(q), torch.distributions.Bernoulli(p))
    expected_grad = grad(kl, [q])[0]
    elbo = TraceEnum_ELBO(max_plate_nesting=1, strict_enumeration_warning=True)
    actual_loss = elbo.loss_and_grads(model, guide)
    assert abs(actual_loss - kl.item()) < 0.05 * kl.item(), (actual_loss, kl.item())
    assert abs(float(pyro.param("q").grad) - float(expected_grad)) < 0.05 * abs(float(expected_grad)), \
        (float(pyro.param("q").grad), float(expected_grad))
    # shapes: the draws sit on their own dim left of the plate
    tr = poutine.trace(poutine.enum(guide, first_available_dim=-2)).get_trace()
    assert tr.nodes["y"]["value"].shape == (num_samples, 1)
    assert tr.nodes["z"]["value"].shape[-1] == 3 and tr.nodes["z"]["value"].shape[0] == num_samples


def run_local_sampling_of_a_reparameterised_site(device):
    """A Normal guide site with num_samples: pathwise gradients, each draw weighing 1/n."""
    pyro.clear_param_store()
    pyro.set_rng_seed(1)
    loc = pyro.param("loc", torch.tensor(0.3, dtype=torch.float64, device=device, requires_grad=True))
    one = torch.ones((), dtype=torch.float64, device=device)

    def model():
        pyro.sample("x", dist.Normal(0.0 * one, one))

    def guide():
        pyro.sample("x", dist.Normal(pyro.param("loc"), one),
                    infer={"enumerate": "parallel", "num_samples": 50000})

    elbo = TraceEnum_ELBO(max_plate_nesting=0)
    loss = elbo.loss_and_grads(model, guide)
    # KL(N(loc,1) || N(0,1)) = loc^2 / 2, gradient loc
    assert abs(loss - 0.045) < 0.01, loss
    assert abs(float(pyro.param("loc").grad) - 0.3) < 0.02
