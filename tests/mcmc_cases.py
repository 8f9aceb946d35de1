"""MCMC test bodies shared by the CPU suite (kernels answered by the numpy oracle through
tests/oracle_backend.py: exercises the HOST driver only) and the GPU suite (real HIP kernels)."""
import math
import os

import numpy as np
import torch

import pyro_amd as pyro
import pyro_amd.distributions as dist
from oracle import integrator as o_int
from oracle import nuts as o_nuts
from pyro_amd.infer.mcmc import HMC, MCMC, NUTS, GaussianPotential
from pyro_amd.ops.integrator import velocity_verlet

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"), allow_pickle=False)


def make_precision(D, seed):
    g = np.random.default_rng(seed)
    A = g.standard_normal((D, D))
    Lam = np.linalg.inv(A @ A.T / D + 0.1 * np.eye(D))
    return 0.5 * (Lam + Lam.T)


def run_integrator_golden(device, rtol):
    """pyro_amd.ops.integrator.velocity_verlet on the reference's own outputs
    (tests/golden/integrator.npz, generated from pyro/ops/integrator.py)."""
    g = load("integrator")
    for name, pot in (("harmonic", lambda q: 0.5 * q["x"] ** 2),
                      ("quartic", lambda q: 0.25 * q["x"].pow(4))):
        z = {"x": torch.tensor([float(g[name + "/q0"])], dtype=torch.float64, device=device)}
        r = {"x": torch.tensor([float(g[name + "/p0"])], dtype=torch.float64, device=device)}

        class KG:
            def __call__(self, p):
                return {"x": p["x"]}

            def inverse_mass_diag(self, site):
                return torch.ones(1, dtype=torch.float64, device=device)

        zf, rf, gf, pe = velocity_verlet(z, r, lambda q: pot(q).sum(), KG(), float(g[name + "/eps"]),
                                         int(g[name + "/n"]))
        np.testing.assert_allclose(zf["x"].item(), float(g[name + "/qf"]), rtol=rtol, atol=rtol)
        np.testing.assert_allclose(rf["x"].item(), float(g[name + "/pf"]), rtol=rtol, atol=rtol)
        np.testing.assert_allclose(pe.item(), float(g[name + "/pe"]), rtol=rtol, atol=rtol)
    Lt = torch.tensor(g["gauss/Lambda"], device=device)
    im = torch.tensor(g["gauss/inv_mass"], device=device)

    class KG2:
        def __call__(self, p):
            return {"x": im * p["x"]}

        def inverse_mass_diag(self, site):
            return im

    z = {"x": torch.tensor(g["gauss/z0"], device=device)}
    r = {"x": torch.tensor(g["gauss/r0"], device=device)}
    zf, rf, gf, pe = velocity_verlet(z, r, lambda q: 0.5 * q["x"] @ Lt @ q["x"], KG2(),
                                     float(g["gauss/eps"]), int(g["gauss/n"]))
    np.testing.assert_allclose(zf["x"].cpu().numpy(), g["gauss/zf"], rtol=rtol, atol=rtol)
    np.testing.assert_allclose(rf["x"].cpu().numpy(), g["gauss/rf"], rtol=rtol, atol=rtol)
    np.testing.assert_allclose(gf["x"].cpu().numpy(), g["gauss/gf"], rtol=rtol, atol=rtol)
    np.testing.assert_allclose(pe.item(), float(g["gauss/pe"]), rtol=rtol)
    # the input dicts are not modified (integrator.py:36-37 copies them)
    np.testing.assert_array_equal(z["x"].cpu().numpy(), g["gauss/z0"])


def _np_potentials(kind, Lam):
    if kind == "gaussian":
        return o_int.gaussian_potential(Lam)

    def fn(z):   # non-quadratic: 0.5 z'Lz + sum log cosh z
        g = Lam @ z
        return 0.5 * float(z @ g) + float(np.sum(np.logaddexp(z, -z) - math.log(2.0))), \
            g + np.tanh(z)
    return fn


class LogCoshPotential:
    def __init__(self, Lam):
        self.L = Lam

    def __call__(self, z):
        x = z["x"]
        g = x @ self.L
        return 0.5 * (x * g).sum(-1) + (torch.logaddexp(x, -x) - math.log(2.0)).sum(-1)


def run_nuts_chains_vs_oracle(device, D, C, kind, multinomial, n_trans, fused, rtol=1e-8,
                              max_tree_depth=6, dtype=torch.float64, async_chains=None, min_slots=None):
    """NUTS kernel (no adaptation) driven through MCMC, chain-for-chain against the recursive
    restatement of the reference with the same keyed draws.  ``async_chains`` True / False pins the
    schedule (spans of asynchronous chains: pa_nuts_tree_run_begin/_advance, captured rounds; or the
    lock-step per-transition path) and asserts it ran; ``min_slots`` switches compacted rounds on."""
    Lam = make_precision(D, 11)
    g = np.random.default_rng(5)
    z0 = g.standard_normal((C, D)) * 0.4
    inv_mass = g.uniform(0.5, 1.5, (C, D))
    steps = g.uniform(0.08, 0.3, C)
    Lt = torch.tensor(Lam, dtype=dtype, device=device)
    pot = GaussianPotential(Lt) if kind == "gaussian" else LogCoshPotential(Lt)
    pyro.set_rng_seed(123)
    kernel = NUTS(potential_fn=pot, step_size=1.0, adapt_step_size=False, adapt_mass_matrix=False,
                  use_multinomial_sampling=multinomial, max_tree_depth=max_tree_depth)
    kernel.use_fused_gaussian = fused
    if async_chains is not None:
        kernel.use_async_chains = async_chains
        kernel.compact_chains = min_slots is not None
        if min_slots is not None:
            kernel.min_slots, kernel.sync_every, kernel.rounds_per_replay = min_slots, 2, 4
    mcmc = MCMC(kernel, num_samples=n_trans, warmup_steps=0, num_chains=C,
                initial_params={"x": torch.tensor(z0, dtype=dtype, device=device)})
    # fixed per-chain step sizes / masses: set after setup through a hook on the first call
    orig_setup = kernel.setup

    def setup(warmup_steps, *a, **k):
        orig_setup(warmup_steps, *a, **k)
        kernel._adapter.step_size = torch.tensor(steps, dtype=dtype, device=device)
        kernel.mass_matrix_adapter.inverse_mass_matrix = torch.tensor(inv_mass, dtype=dtype,
                                                                     device=device)
    kernel.setup = setup
    mcmc.run()
    if async_chains is not None:
        assert kernel.bulk_ready == async_chains
        assert (kernel._span_rounds > 0) == async_chains, "the asynchronous schedule did not run"
        if async_chains and device.type == "cuda" and n_trans >= 8:
            assert kernel._span_replays > 0, "no round of the span was a captured graph"
        if async_chains and min_slots is not None:
            assert kernel._span_compactions > 0, "no round was compacted"
    samples = mcmc.get_samples(group_by_chain=True)["x"].cpu().numpy()   # [C, S, D]
    assert samples.shape == (C, n_trans, D)
    pg = _np_potentials(kind, Lam)
    np_dt = np.float64 if dtype == torch.float64 else np.float32
    nleap = 0
    for c in range(C):
        z = z0[c].copy()
        pe, gr = pg(z)
        for t in range(n_trans):
            out = o_nuts.nuts_transition(z, pe, gr, pg, inv_mass[c], steps[c],
                                         o_nuts.KeyedDraws(123, c, t, np_dt), max_tree_depth,
                                         multinomial, dtype=np_dt)
            z, pe, gr = out["z"], out["pe"], out["grad"]
            nleap += out["n_leapfrog"]
            np.testing.assert_allclose(samples[c, t], z, rtol=rtol, atol=rtol,
                                       err_msg="chain %d transition %d" % (c, t))
    assert kernel.num_leapfrog_steps == nleap
    return samples


def make_dense_inverse_mass(C, D, seed):
    g = np.random.default_rng(seed)
    out = []
    for _ in range(C):
        B = g.standard_normal((D, D))
        out.append(B @ B.T / D + 0.3 * np.eye(D))
    return np.stack(out)


def run_nuts_dense_mass_vs_oracle(device, D, C, kind, multinomial, n_trans, rtol=1e-8,
                                  max_tree_depth=5, dtype=torch.float64):
    """NUTS(full_mass=True) with a different dense inverse mass per chain, chain-for-chain against
    the restatement of the reference (which is pinned on reference runs with full_mass=True,
    tests/golden/nuts_reference.npz d10_dense / d12_dense_slice).  The product runs in whitened
    coordinates; the oracle runs the reference's formulation."""
    Lam = make_precision(D, 21)
    g = np.random.default_rng(6)
    z0 = g.standard_normal((C, D)) * 0.4
    inv_mass = make_dense_inverse_mass(C, D, 7)
    steps = g.uniform(0.1, 0.35, C)
    Lt = torch.tensor(Lam, dtype=dtype, device=device)
    pot = GaussianPotential(Lt) if kind == "gaussian" else LogCoshPotential(Lt)
    pyro.set_rng_seed(321)
    kernel = NUTS(potential_fn=pot, step_size=1.0, adapt_step_size=False, adapt_mass_matrix=False,
                  use_multinomial_sampling=multinomial, max_tree_depth=max_tree_depth,
                  full_mass=True)
    mcmc = MCMC(kernel, num_samples=n_trans, warmup_steps=0, num_chains=C,
                initial_params={"x": torch.tensor(z0, dtype=dtype, device=device)})
    orig_setup = kernel.setup

    def setup(warmup_steps, *a, **k):
        orig_setup(warmup_steps, *a, **k)
        kernel._adapter.step_size = torch.tensor(steps, dtype=dtype, device=device)
        kernel.mass_matrix_adapter.inverse_mass_matrix = torch.tensor(inv_mass, dtype=dtype,
                                                                     device=device)
    kernel.setup = setup
    mcmc.run()
    assert not kernel._fused
    samples = mcmc.get_samples(group_by_chain=True)["x"].cpu().numpy()
    pg = _np_potentials(kind, Lam)
    np_dt = np.float64 if dtype == torch.float64 else np.float32
    nleap = 0
    for c in range(C):
        z = z0[c].copy()
        pe, gr = pg(z)
        for t in range(n_trans):
            out = o_nuts.nuts_transition(z, pe, gr, pg, inv_mass[c], steps[c],
                                         o_nuts.KeyedDraws(321, c, t, np_dt), max_tree_depth,
                                         multinomial, dtype=np_dt)
            z, pe, gr = out["z"], out["pe"], out["grad"]
            nleap += out["n_leapfrog"]
            np.testing.assert_allclose(samples[c, t], z, rtol=rtol, atol=rtol,
                                       err_msg="chain %d transition %d" % (c, t))
    assert kernel.num_leapfrog_steps == nleap


def run_dense_mass_products_vs_reference(device):
    """DenseMassMatrix against BlockMassMatrix of the reference (tests/golden/adaptation.npz
    block/*): one dense block over sites (a, b), site c diagonal; Welford adaptation, then the
    three products."""
    from pyro_amd.infer.mcmc.adaptation import DenseMassMatrix, block_mask
    from pyro_amd.infer.mcmc.util import Layout
    g = load("adaptation")
    layout = Layout({"a": (3,), "b": (), "c": (2,)})
    mask = block_mask(layout, [("a", "b")])
    C = 2                                    # second chain sees the samples scaled by 2
    mm = DenseMassMatrix(C, 6, torch.float64, device, mask=mask)
    for z in g["block/zs"]:
        mm.update(torch.tensor(np.stack([z, 2 * z]), device=device))
    mm.end_adaptation()
    V = mm.inverse_mass_matrix.cpu().numpy()
    np.testing.assert_allclose(V[0][:4, :4], g["block/inv_ab"], rtol=1e-10)
    np.testing.assert_allclose(np.diag(V[0])[4:], g["block/inv_c"], rtol=1e-10)
    assert np.all(V[0][:4, 4:] == 0) and V[0][4, 5] == 0
    # chain 1 differs only through the regulariser's additive shrinkage term
    n = len(g["block/zs"])
    shrink = 1e-3 * (5.0 / (n + 5.0))
    np.testing.assert_allclose(V[1][:4, :4] - shrink * np.eye(4),
                               4 * (g["block/inv_ab"] - shrink * np.eye(4)), rtol=1e-9, atol=1e-12)
    r = torch.tensor(np.stack([g["block/r"], g["block/r"]]), device=device)
    np.testing.assert_allclose(mm.kinetic_grad(r)[0].cpu().numpy(), g["block/kinetic_grad"], rtol=1e-10)
    np.testing.assert_allclose(mm.scale(r)[0].cpu().numpy(), g["block/scale"], rtol=1e-10)
    np.testing.assert_allclose(mm.unscale(r)[0].cpu().numpy(), g["block/unscale"], rtol=1e-10)
    # whitened coordinates: color(whiten(z)) = z, and unit kinetic energy of r' = L^T r
    z = torch.tensor(g["block/zs"][:2].copy(), device=device)
    np.testing.assert_allclose(mm.color(mm.whiten(z)).cpu().numpy(), z.cpu().numpy(), rtol=1e-10)
    ru = mm.unscale(r)
    np.testing.assert_allclose((ru * ru).sum(-1).cpu().numpy(),
                               (r * mm.kinetic_grad(r)).sum(-1).cpu().numpy(), rtol=1e-10)


def run_dense_mass_adaptation(device, dtype, C=4, D=5, warmup=150, S=150, check=True):
    """Warm-up with dense mass adaptation on a correlated Gaussian, NUTS and HMC: the adapted
    inverse mass approaches the target covariance and the draws have the right moments."""
    Lam = make_precision(D, 31)
    Sigma = np.linalg.inv(Lam)
    Lt = torch.tensor(Lam, dtype=dtype, device=device)
    out = {}
    for name, kernel in (("nuts", NUTS(potential_fn=GaussianPotential(Lt), full_mass=True,
                                        max_tree_depth=5)),
                         ("hmc", HMC(potential_fn=GaussianPotential(Lt), full_mass=True,
                                     trajectory_length=1.5))):
        pyro.set_rng_seed(7)
        g = np.random.default_rng(8)
        init = {"x": torch.tensor(g.standard_normal((C, D)), dtype=dtype, device=device)}
        mcmc = MCMC(kernel, num_samples=S, warmup_steps=warmup, num_chains=C, initial_params=init)
        mcmc.run()
        V = kernel.mass_matrix_adapter.inverse_mass_matrix.cpu().numpy()
        x = mcmc.get_samples()["x"].cpu().numpy().astype(np.float64)
        out[name] = (V, x)
        if check:
            assert V.shape == (C, D, D)
            off = np.abs(V - np.transpose(V, (0, 2, 1))).max()
            assert off < 1e-6
            # dense adaptation picked up the correlations: closer to Sigma than its own diagonal
            err_dense = np.abs(V.mean(0) - Sigma).max()
            err_diag = np.abs(np.diag(np.diag(Sigma)) - Sigma).max()
            assert err_dense < 0.6 * err_diag, (name, err_dense, err_diag)
            emp = np.cov(x.T)
            assert np.abs(x.mean(0)).max() < 0.5 * np.sqrt(np.diag(Sigma)).max()
            assert np.abs(emp - Sigma).max() < 0.5 * np.abs(Sigma).max(), name
    return out


def run_structured_mass(device, dtype=torch.float32, warmup=1000, C=4):
    """tests/infer/mcmc/test_nuts.py:465-503 test_structured_mass: full_mass False / True give a
    diagonal / dense inverse mass over all sites; full_mass=[("w",), ("x", "y")] adapts two dense
    blocks and a diagonal rest, each approaching the corresponding block of the target covariance
    (atol = rtol = 0.5 as in the reference).  Chains are vectorised here, so the sites are
    concatenated along the last dim."""
    def t(v):
        return torch.tensor(v, dtype=dtype, device=device)

    def model(cov):
        def wide(n):
            return dist.Normal(torch.zeros(n, dtype=dtype, device=device), 1000.0).to_event(1)
        w = pyro.sample("w", wide(2))
        x = pyro.sample("x", wide(1))
        y = pyro.sample("y", wide(1))
        z = pyro.sample("z", wide(1))
        wxyz = torch.cat([w, x, y, z], dim=-1)
        pyro.sample("obs", dist.MultivariateNormal(torch.zeros(5, dtype=dtype, device=device), cov),
                    obs=wxyz)

    w_cov, xy_cov, z_var = t([[1.5, 0.5], [0.5, 1.5]]), t([[2.0, 1.0], [1.0, 3.0]]), t([2.5])
    cov = torch.zeros(5, 5, dtype=dtype, device=device)
    cov[:2, :2], cov[2:4, 2:4], cov[4, 4] = w_cov, xy_cov, z_var[0]
    for dense_mass in (True, False):                                      # smoke tests
        pyro.set_rng_seed(0)
        kernel = NUTS(model, full_mass=dense_mass, max_tree_depth=4)
        MCMC(kernel, num_samples=1, warmup_steps=1, num_chains=2).run(cov)
        assert kernel.mass_matrix_adapter.inverse_mass_matrix.dim() == 2 + int(dense_mass)   # chain dim
        assert kernel.inverse_mass_matrix[("w", "x", "y", "z")].dim() == 1 + 1 + int(dense_mass)
    pyro.set_rng_seed(1)
    kernel = NUTS(model, full_mass=[("w",), ("x", "y")], max_tree_depth=6)
    mcmc = MCMC(kernel, num_samples=1, warmup_steps=warmup, num_chains=C)
    mcmc.run(cov)
    V = kernel.mass_matrix_adapter.inverse_mass_matrix                     # [C, 5, 5]
    # the reference's public form (tests/infer/mcmc/test_nuts.py:501-503): one entry per block
    blocks = kernel.inverse_mass_matrix
    assert set(blocks) == {("w",), ("x", "y"), ("z",)}
    lead = (C,) if C > 1 else ()
    assert blocks[("w",)].shape == lead + (2, 2) and blocks[("x", "y")].shape == lead + (2, 2)
    assert blocks[("z",)].shape == lead + (1,)
    first = (lambda t: t[0]) if C > 1 else (lambda t: t)
    torch.testing.assert_close(first(blocks[("w",)]), w_cov, atol=0.5, rtol=0.5)
    torch.testing.assert_close(first(blocks[("x", "y")]), xy_cov, atol=0.5, rtol=0.5)
    torch.testing.assert_close(first(blocks[("z",)]), z_var, atol=0.5, rtol=0.5)
    sl = kernel._layout.slices
    assert list(kernel._layout.names) == ["w", "x", "y", "z"]
    w0, w1 = sl["w"]
    x0, y1 = sl["x"][0], sl["y"][1]
    z0 = sl["z"][0]
    for c in range(C):
        torch.testing.assert_close(V[c, w0:w1, w0:w1], w_cov, atol=0.5, rtol=0.5)
        torch.testing.assert_close(V[c, x0:y1, x0:y1], xy_cov, atol=0.5, rtol=0.5)
        torch.testing.assert_close(V[c, z0, z0].reshape(1), z_var, atol=0.5, rtol=0.5)
        # the structure: nothing outside the two dense blocks and the diagonal
        off = V[c].clone()
        off[w0:w1, w0:w1] = 0
        off[x0:y1, x0:y1] = 0
        off[z0, z0] = 0
        assert float(off.abs().max()) == 0.0


def logreg_mcmc_model(X, y):
    D = X.shape[1]
    w = pyro.sample("w", dist.Normal(torch.zeros(D, dtype=X.dtype, device=X.device),
                                     torch.ones(D, dtype=X.dtype, device=X.device)).to_event(1))
    with pyro.plate("data", X.shape[0]):
        pyro.sample("obs", dist.Bernoulli(logits=dist.linear_logits(X, w)), obs=y)


def logreg_mcmc_model_unfused(X, y):
    D = X.shape[1]
    w = pyro.sample("w", dist.Normal(torch.zeros(D, dtype=X.dtype, device=X.device),
                                     torch.ones(D, dtype=X.dtype, device=X.device)).to_event(1))
    with pyro.plate("data", X.shape[0]):
        logits = w @ X.t()
        logits = logits.squeeze(-2) if logits.dim() > 1 else logits
        pyro.sample("obs", dist.Bernoulli(logits=logits), obs=y)


def run_persistent_equals_stepwise(device, dtype, rtol, C=6, D=9, warmup=40, S=6):
    """MCMC(NUTS(GaussianPotential)) with step-size + mass adaptation: the persistent multi-transition
    launches (in-kernel dual averaging / Welford, host window ends) against the per-transition
    host-adapted path.  Same Philox keys => the same chains up to rounding of the adaptation math."""
    Lam = torch.tensor(make_precision(D, 4), dtype=dtype, device=device)
    z0 = torch.tensor(np.random.default_rng(1).standard_normal((C, D)) * 0.3, dtype=dtype,
                      device=device)
    outs = []
    for persistent in (False, True):
        pyro.set_rng_seed(77)
        kernel = NUTS(potential_fn=GaussianPotential(Lam), max_tree_depth=5)
        kernel.use_persistent = persistent
        mcmc = MCMC(kernel, num_samples=S, warmup_steps=warmup, num_chains=C,
                    initial_params={"x": z0.clone()})
        mcmc.run()
        outs.append((mcmc.get_samples(group_by_chain=True)["x"].clone(),
                     kernel.step_size.clone(), kernel.mass_matrix_adapter.inverse_mass_matrix.clone(),
                     kernel.num_leapfrog_steps, mcmc.diagnostics()))
    a, b = outs
    assert a[3] == b[3], (a[3], b[3])                       # identical trees
    torch.testing.assert_close(a[1], b[1], rtol=rtol, atol=0)
    torch.testing.assert_close(a[2], b[2], rtol=rtol, atol=0)
    torch.testing.assert_close(a[0], b[0], rtol=rtol, atol=rtol)
    assert a[4]["acceptance rate"] == b[4]["acceptance rate"]
    assert a[4]["divergences"] == b[4]["divergences"]
    assert abs(a[4]["mean tree depth"] - b[4]["mean tree depth"]) < 1e-12


def run_async_equals_lockstep(device, dtype, rtol, C=6, D=9, warmup=40, S=6, multinomial=True, adapt=True,
                              min_slots=None):
    """MCMC(NUTS(generic potential)) with step-size + mass adaptation: spans of ASYNCHRONOUS chains
    (a chain that finishes a tree adapts and starts its next tree in the same launch, in-kernel dual
    averaging / Welford) against the lock-step per-transition path with host adaptation.  The Philox
    keys do not know the schedule => the same chains up to rounding of the adaptation math."""
    from pyro_amd.ops import fuser
    Lam = torch.tensor(make_precision(D, 4), dtype=dtype, device=device)
    z0 = torch.tensor(np.random.default_rng(1).standard_normal((C, D)) * 0.3, dtype=dtype,
                      device=device)
    outs = []
    # the two schedules are compared on the SAME potential kernels: the captured rounds of the asynchronous
    # path would otherwise run the potential's element-wise glue as generated kernels (sums in another
    # order), the eager lock-step rounds as ATen's
    fused_glue, fuser.ENABLED["on"] = fuser.ENABLED["on"], False
    for async_chains in (False, True):
        pyro.set_rng_seed(78)
        kernel = NUTS(potential_fn=LogCoshPotential(Lam), max_tree_depth=5, step_size=1.0 if adapt else 0.15,
                      use_multinomial_sampling=multinomial, adapt_step_size=adapt, adapt_mass_matrix=adapt)
        kernel.use_async_chains = async_chains
        kernel.compact_chains = min_slots is not None      # late in a span: rounds over the active chains only
        if min_slots is not None:
            kernel.min_slots, kernel.sync_every, kernel.rounds_per_replay = min_slots, 2, 4
        mcmc = MCMC(kernel, num_samples=S, warmup_steps=warmup, num_chains=C,
                    initial_params={"x": z0.clone()})
        mcmc.run()
        assert kernel.bulk_ready == async_chains
        if async_chains and min_slots is not None:
            assert kernel._span_compactions > 0, "no round was compacted"
        outs.append((mcmc.get_samples(group_by_chain=True)["x"].clone(),
                     kernel.step_size.clone(), kernel.mass_matrix_adapter.inverse_mass_matrix.clone(),
                     kernel.num_leapfrog_steps, mcmc.diagnostics(), kernel._mean_accept_prob.clone()))
    fuser.ENABLED["on"] = fused_glue
    a, b = outs
    assert a[3] == b[3], (a[3], b[3])                       # identical trees
    if not adapt and min_slots is None:
        # no host-side adaptation arithmetic to differ from the kernel's: the two schedules run the same
        # kernels on the same numbers
        rtol = 0.0
    torch.testing.assert_close(a[1], b[1], rtol=rtol, atol=0)
    torch.testing.assert_close(a[2], b[2], rtol=rtol, atol=0)
    torch.testing.assert_close(a[0], b[0], rtol=rtol, atol=rtol)
    torch.testing.assert_close(a[5], b[5], rtol=max(rtol, 1e-14), atol=max(rtol, 1e-14))
    assert a[4]["acceptance rate"] == b[4]["acceptance rate"]
    assert a[4]["divergences"] == b[4]["divergences"]
    assert abs(a[4]["mean tree depth"] - b[4]["mean tree depth"]) < 1e-12


# ---- discrete latents summed out of the potential (tests/golden/mcmc_enum.npz) -------------------
def _enum_models(device, dtype, batch_safe):
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    from pyro_amd.ops.indexing import Vindex

    def t(v):
        return torch.as_tensor(v, dtype=dtype, device=device)

    K, dim = 3, 3

    def gmm(data):
        phi = pyro.sample("phi", dist.Dirichlet(torch.ones(K, dtype=dtype, device=device)))
        with pyro.plate("num_clusters", K):
            means = pyro.sample("cluster_means", dist.Normal(t([0.0, 1.0, 2.0]), t(1.0)))
        with pyro.plate("data", data.shape[0]):
            a = pyro.sample("assignments", dist.Categorical(phi))
            # tests/infer/mcmc/test_nuts.py:286 indexes means[a]; with a leading chain dim the
            # cluster dim has to move off the data plate's dim first
            m = Vindex(means.unsqueeze(-2))[..., a] if batch_safe else means[a]
            pyro.sample("obs", dist.Normal(m, t(1.0)), obs=data)

    def bern(data):
        y_prob = pyro.sample("y_prob", dist.Beta(t(1.0), t(1.0)))
        with pyro.plate("data", data.shape[0]):
            y = pyro.sample("y", dist.Bernoulli(y_prob))
            z = pyro.sample("z", dist.Bernoulli(0.65 * y + 0.1))
            pyro.sample("obs", dist.Normal(2.0 * z, t(1.0)), obs=data)

    def hmm(data):
        initialize = pyro.sample("initialize", dist.Dirichlet(torch.ones(dim, dtype=dtype, device=device)))
        with pyro.plate("states", dim):
            transition = pyro.sample("transition",
                                     dist.Dirichlet(torch.ones(dim, dim, dtype=dtype, device=device)))
            loc = pyro.sample("emission_loc", dist.Normal(torch.zeros(dim, dtype=dtype, device=device), t(1.0)))
            scale = pyro.sample("emission_scale",
                                dist.LogNormal(torch.zeros(dim, dtype=dtype, device=device), t(1.0)))
        x = None
        for i, y in pyro.markov(enumerate(data)):
            x = pyro.sample("x_{}".format(i),
                            dist.Categorical(initialize if x is None else transition[x]),
                            infer={"enumerate": "parallel"})
            pyro.sample("y_{}".format(i), dist.Normal(loc[x], scale[x]), obs=y)

    return {"gmm": gmm, "bern": bern, "hmm": hmm}


def run_enum_potential_vs_reference(device, dtype=torch.float64, rtol=1e-9):
    """U(z) and dU/dz with the discrete sites enumerated and summed out equal the reference's
    potential_fn (TraceEinsumEvaluator) at three points per model: one chain at a time through the
    reference-verbatim programs, and all three points as three chains of ONE evaluation."""
    import pyro_amd as pyro
    from pyro_amd.infer.mcmc import initialize_model

    g = load("mcmc_enum")
    for batch_safe, tags in ((False, ("gmm", "bern", "hmm")), (True, ("gmm", "bern"))):
        models = _enum_models(device, dtype, batch_safe)
        for tag in tags:
            data = torch.as_tensor(g[tag + "/data"], dtype=dtype, device=device)
            names = sorted(k.split("/")[2] for k in g.files if k.startswith(tag + "/z0/"))
            pyro.set_rng_seed(0)
            C = 3 if batch_safe else 1
            init, pot, _, _ = initialize_model(models[tag], (data,), max_plate_nesting=1,
                                               num_chains=C)
            assert sorted(init) == names, (sorted(init), names)

            def point(k):
                return {n: torch.as_tensor(g["%s/z%d/%s" % (tag, k, n)], dtype=dtype, device=device)
                        for n in names}

            if batch_safe:
                z = {n: torch.stack([point(k)[n] for k in range(3)]).requires_grad_(True)
                     for n in names}
                from pyro_amd import kernels
                calls, real = [], kernels.mixture_fwd_bwd
                kernels.mixture_fwd_bwd = lambda *a: calls.append(tuple(a[2].shape)) or real(*a)
                try:
                    pe = pot(z)
                finally:
                    kernels.mixture_fwd_bwd = real
                # (three chains of the Gaussian mixture: ONE launch of the mixture leaf kernel, the chains its batch
                #  of parameter sets -- where the kernels serve the data's device)
                if tag == "gmm" and kernels.on_device(data):
                    assert calls == [(3, 3)], calls
                assert pe.shape == (3,)
                grads = torch.autograd.grad(pe.sum(), [z[n] for n in names])
                for k in range(3):
                    np.testing.assert_allclose(pe[k].item(), float(g["%s/pe%d" % (tag, k)]), rtol=rtol)
                    for n, gr in zip(names, grads):
                        ref = g["%s/g%d/%s" % (tag, k, n)]
                        np.testing.assert_allclose(gr[k].cpu().numpy(), ref, rtol=rtol * 100,
                                                   atol=rtol * 100 * float(np.abs(ref).max() + 1e-300))
            else:
                from pyro_amd import kernels
                calls, real = [], kernels.mixture_fwd_bwd
                kernels.mixture_fwd_bwd = lambda *a: calls.append(1) or real(*a)
                try:
                    pot({n: v for n, v in point(0).items()})
                finally:
                    kernels.mixture_fwd_bwd = real
                # (one chain of the Gaussian mixture: the likelihood goes through the mixture leaf kernel)
                # (... where the kernels serve the data's device: the plain-torch host run has no such leaf)
                assert len(calls) == (1 if tag == "gmm" and kernels.on_device(data) else 0), (tag, len(calls))
                for k in range(3):
                    z = {n: v.requires_grad_(True) for n, v in point(k).items()}
                    pe = pot(z)
                    grads = torch.autograd.grad(pe, [z[n] for n in names])
                    np.testing.assert_allclose(pe.item(), float(g["%s/pe%d" % (tag, k)]), rtol=rtol)
                    for n, gr in zip(names, grads):
                        ref = g["%s/g%d/%s" % (tag, k, n)]
                        np.testing.assert_allclose(gr.cpu().numpy(), ref, rtol=rtol * 100,
                                                   atol=rtol * 100 * float(np.abs(ref).max() + 1e-300))


def run_bernoulli_latent_kat(device, dtype=torch.float32, C=4):
    """tests/infer/mcmc/test_nuts.py:306-328 (and test_hmc.py:277-306): posterior mean of y_prob
    with the two Bernoulli layers summed out, here on C chains of one batch."""
    import pyro_amd as pyro
    from pyro_amd.infer.mcmc import MCMC, NUTS

    torch.manual_seed(3)
    N = 2000
    y = (torch.rand(N) < 0.3).to(dtype)
    z = (torch.rand(N) < 0.65 * y + 0.1).to(dtype)
    data = (2.0 * z + torch.randn(N)).to(dtype).to(device)
    pyro.set_rng_seed(0)
    model = _enum_models(device, dtype, True)["bern"]
    mcmc = MCMC(NUTS(model, max_plate_nesting=1), num_samples=150, warmup_steps=100, num_chains=C)
    mcmc.run(data)
    s = mcmc.get_samples()["y_prob"]
    assert s.shape == (150 * C,)
    assert abs(s.mean().item() - 0.3) < 0.05, s.mean().item()


# ---- potentials of models with constrained supports (tests/golden/mcmc_potential.npz) -------------
def _conjugate_models(device, dtype):
    """The conjugate programs of tests/infer/mcmc/test_nuts.py:184-270,394-462 with their batch dims
    declared as plates (same density; needed for the leading chain dim to have a place)."""
    import pyro_amd as pyro
    import pyro_amd.distributions as dist

    def t(v):
        return torch.as_tensor(v, dtype=dtype, device=device)

    def beta_bernoulli(data):
        with pyro.plate("c", 2, dim=-1):
            p = pyro.sample("p_latent", dist.Beta(t([1.1, 1.1]), t([1.1, 1.1])))
            with pyro.plate("d", data.shape[0], dim=-2):
                pyro.sample("obs", dist.Bernoulli(p), obs=data)

    def gamma_normal(data):
        with pyro.plate("c", 2, dim=-1):
            s = pyro.sample("p_latent", dist.Gamma(t([1.0, 1.0]), t([1.0, 1.0])))
            with pyro.plate("d", data.shape[0], dim=-2):
                pyro.sample("obs", dist.Normal(t(3.0), s), obs=data)

    def dirichlet_categorical(data):
        p = pyro.sample("p_latent", dist.Dirichlet(t([1.0, 1.0, 1.0])))
        with pyro.plate("d", data.shape[0]):
            pyro.sample("obs", dist.Categorical(p), obs=data)

    def gamma_beta(data):
        a = pyro.sample("alpha", dist.Gamma(t(1.0), t(1.0)))
        b = pyro.sample("beta", dist.Gamma(t(1.0), t(1.0)))
        with pyro.plate("d", data.shape[0]):
            pyro.sample("x", dist.Beta(a, b), obs=data)

    def beta_binomial(data):
        a = pyro.sample("alpha", dist.HalfCauchy(t(1.0)))
        b = pyro.sample("beta", dist.HalfCauchy(t(1.0)))
        with pyro.plate("plate_0", data.shape[-1]):
            probs = pyro.sample("probs", dist.Beta(a, b))
            with pyro.plate("data", data.shape[0]):
                pyro.sample("binomial", dist.Binomial(probs=probs, total_count=1000), obs=data)

    def gamma_poisson(data):
        a = pyro.sample("alpha", dist.HalfCauchy(t(1.0)))
        b = pyro.sample("beta", dist.HalfCauchy(t(1.0)))
        with pyro.plate("plate_0", data.shape[-1]):
            rate = pyro.sample("rate", dist.Gamma(a, b))
            with pyro.plate("data", data.shape[0]):
                pyro.sample("obs", dist.Poisson(rate), obs=data)

    return dict(beta_bernoulli=beta_bernoulli, gamma_normal=gamma_normal,
                dirichlet_categorical=dirichlet_categorical, gamma_beta=gamma_beta,
                beta_binomial=beta_binomial, gamma_poisson=gamma_poisson)


def run_constrained_potentials_vs_reference(device, dtype=torch.float64, rtol=1e-9):
    """U(z) = -log p(T^-1(z), data) - log|det J| and its gradient equal the reference's
    potential_fn: per chain (unbatched call) and for three chains in one evaluation."""
    import pyro_amd as pyro
    from pyro_amd.infer.mcmc import initialize_model

    g = load("mcmc_potential")
    models = _conjugate_models(device, dtype)
    for tag, model in models.items():
        raw = g[tag + "/data"]
        data = torch.as_tensor(raw, device=device)
        data = data.to(dtype) if tag != "dirichlet_categorical" else data.long()
        names = sorted(k.split("/")[2] for k in g.files if k.startswith(tag + "/z0/"))

        def point(k):
            return {n: torch.as_tensor(g["%s/z%d/%s" % (tag, k, n)], dtype=dtype, device=device)
                    for n in names}

        def check(pe, grads, k, sel):
            np.testing.assert_allclose(sel(pe).item(), float(g["%s/pe%d" % (tag, k)]), rtol=rtol,
                                       err_msg=tag)
            for n, gr in zip(names, grads):
                ref = g["%s/g%d/%s" % (tag, k, n)]
                np.testing.assert_allclose(sel(gr).cpu().numpy(), ref, rtol=rtol * 100,
                                           atol=rtol * 100 * float(np.abs(ref).max() + 1e-300),
                                           err_msg=tag + "/" + n)

        for C in (1, 3):
            pyro.set_rng_seed(0)
            init, pot, _, _ = initialize_model(model, (data,), num_chains=C)
            assert sorted(init) == names, (sorted(init), names)
            if C == 1:
                for k in range(3):
                    z = {n: v.requires_grad_(True) for n, v in point(k).items()}
                    pe = pot(z)
                    check(pe, torch.autograd.grad(pe, [z[n] for n in names]), k, lambda x: x)
            else:
                z = {n: torch.stack([point(k)[n] for k in range(3)]).requires_grad_(True)
                     for n in names}
                pe = pot(z)
                assert pe.shape == (3,)
                grads = torch.autograd.grad(pe.sum(), [z[n] for n in names])
                for k in range(3):
                    check(pe, grads, k, lambda x, k=k: x[k])


def run_sequential_consistent(device, dtype=torch.float64):
    """tests/infer/mcmc/test_mcmc_api.py:326-368 (nothing left over from a previous run) in this
    backend's terms: a kernel object that already ran -- adapted step size, mass matrix, tree
    buffers, graphs -- reproduces its chains from the same seed, and equals a fresh kernel.  (Chain
    c of a batch equals the chain a run restricted to it produces: test_chain_offset_shifts_streams
    on the GPU and the chain-sharded gloo test.)"""
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    from pyro_amd.infer.mcmc import MCMC, NUTS

    data = torch.tensor([1.0], dtype=dtype, device=device)

    def model(data):
        x = pyro.sample("x", dist.Normal(torch.zeros((), dtype=dtype, device=device), 1.0))
        y = pyro.sample("y", dist.Normal(x, 1.0))
        with pyro.plate("d", 1):
            pyro.sample("obs", dist.Normal(y, 1.0), obs=data)

    def run(kernel):
        pyro.set_rng_seed(7)
        z = torch.linspace(-1.0, 1.0, 6, dtype=dtype, device=device).reshape(3, 2)
        m = MCMC(kernel, num_samples=30, warmup_steps=30, num_chains=3,
                 initial_params={"x": z[:, 0].clone(), "y": z[:, 1].clone()})
        m.run(data)
        s = m.get_samples(group_by_chain=True)
        return np.stack([s["x"].cpu().numpy(), s["y"].cpu().numpy()])

    kernel = NUTS(model, max_tree_depth=5)
    first = run(kernel)
    assert first.shape == (2, 3, 30) and np.isfinite(first).all()
    assert np.abs(first[:, 0] - first[:, 1]).max() > 0        # chains differ from each other
    np.testing.assert_array_equal(run(kernel), first)          # the same object, second run
    np.testing.assert_array_equal(run(NUTS(model, max_tree_depth=5)), first)


# ---- conjugate Gaussian chains (tests/infer/mcmc/test_nuts.py:30-146, test_hmc.py:34-170) ----------
GAUSSIAN_CHAINS = {
    # id: (dim, chain_len, num_obs, num_samples, expected_means, expected_precs, mean_tol, std_tol)
    "dim=10_chain-len=3_num_obs=1": (10, 3, 1, 800, [0.25, 0.50, 0.75], [1.33, 1, 1.33], 0.09, 0.09),
    "dim=10_chain-len=4_num_obs=1": (10, 4, 1, 1600, [0.20, 0.40, 0.60, 0.80],
                                     [1.25, 0.83, 0.83, 1.25], 0.07, 0.06),
    "dim=5_chain-len=2_num_obs=10000": (5, 2, 10000, 800, [0.5, 1.0], [2.0, 10000], 0.05, 0.05),
    "dim=5_chain-len=9_num_obs=1": (5, 9, 1, 1400, [0.10, 0.20, 0.30, 0.40, 0.50, 0.60, 0.70, 0.80, 0.90],
                                    [1.11, 0.63, 0.48, 0.42, 0.4, 0.42, 0.48, 0.63, 1.11], 0.08, 0.08),
}


def run_gaussian_chain(device, case, kernel="nuts", dtype=torch.float32, C=8, **kernel_kwargs):
    """The reference's conjugate Gaussian-chain sampler test: loc_1 ~ N(0,1), loc_i ~ N(loc_{i-1},1),
    obs ~ N(loc_T, 1); rmse of the posterior means / stds of every loc_i against the closed form.
    The reference's ``num_samples`` are spread over C chains of one batch (200 warm-up steps each)."""
    import pyro_amd as pyro
    import pyro_amd.distributions as dist
    from pyro_amd.infer.mcmc import HMC, MCMC, NUTS

    dim, chain_len, num_obs, num_samples, means, precs, mean_tol, std_tol = GAUSSIAN_CHAINS[case]
    data = torch.ones(num_obs, dim, dtype=dtype, device=device)
    one = torch.ones((), dtype=dtype, device=device)

    def model(data):
        loc = torch.zeros(dim, dtype=dtype, device=device)
        with pyro.plate("dim", dim, dim=-1):
            for i in range(1, chain_len + 1):
                loc = pyro.sample("loc_{}".format(i), dist.Normal(loc, one))
            with pyro.plate("obs_plate", num_obs, dim=-2):
                pyro.sample("obs", dist.Normal(loc, one), obs=data)

    pyro.set_rng_seed(0)
    k = NUTS(model, **kernel_kwargs) if kernel == "nuts" else HMC(model, **kernel_kwargs)
    mcmc = MCMC(k, num_samples=max(num_samples // C, 50) * 2, warmup_steps=200, num_chains=C)
    mcmc.run(data)
    samples = mcmc.get_samples()
    for i in range(1, chain_len + 1):
        latent = samples["loc_{}".format(i)].double()
        exp_mean = torch.full((dim,), means[i - 1], dtype=torch.float64, device=device)
        exp_std = 1.0 / torch.sqrt(torch.full((dim,), float(precs[i - 1]), dtype=torch.float64,
                                              device=device))
        e_mean = (latent.mean(0) - exp_mean).pow(2).mean().sqrt().item()
        e_std = (latent.std(0) - exp_std).pow(2).mean().sqrt().item()
        assert e_mean < mean_tol and e_std < std_tol, (case, i, e_mean, e_std)


def run_arrowhead_mass(device, dtype=torch.float32, warmup=1000, C=4):
    """tests/infer/mcmc/test_nuts.py:506-546 test_arrowhead_mass: NUTS with
    ``kernel.mass_matrix_adapter = ArrowheadMassMatrix()`` and full_mass=[("w",), ("y", "x")] adapts
    an arrowhead MASS matrix (head = w, y, x; tail = z diagonal) from the potential's gradients; its
    head rows and tail diagonal approach those of the target precision (atol = rtol = 0.2 as in the
    reference, per chain)."""
    from pyro_amd.infer.mcmc import ArrowheadMassMatrix

    def model(prec):
        def wide(n):
            return dist.Normal(torch.zeros(n, dtype=dtype, device=device), 1000.0).to_event(1)
        w = pyro.sample("w", wide(2))
        x = pyro.sample("x", wide(1))
        y = pyro.sample("y", wide(1))
        z = pyro.sample("z", wide(2))
        wyxz = torch.cat([w, y, x, z], dim=-1)
        pyro.sample("obs", dist.MultivariateNormal(torch.zeros(6, dtype=dtype, device=device),
                                                   precision_matrix=prec), obs=wyxz)

    g = torch.Generator().manual_seed(5)
    A = torch.randn(6, 12, generator=g, dtype=torch.float64)
    prec = (A @ A.t() * 0.1).to(dtype=dtype, device=device)
    pyro.set_rng_seed(2)
    kernel = NUTS(model, full_mass=[("w",), ("y", "x")], max_tree_depth=7)
    kernel.mass_matrix_adapter = ArrowheadMassMatrix()
    mcmc = MCMC(kernel, num_samples=1, warmup_steps=warmup, num_chains=C)
    mcmc.run(prec)
    assert ("w", "y", "x", "z") in kernel.inverse_mass_matrix
    keyed = kernel.mass_matrix_adapter.mass_matrix[("w", "y", "x", "z")]       # the reference's access
    assert tuple(keyed.top.shape)[-2:] == (4, 6) and tuple(keyed.bottom_diag.shape)[-1:] == (2,)
    top, bottom = kernel.mass_matrix_adapter.mass_matrix       # reference order: w, y, x | z
    assert tuple(top.shape) == (C, 4, 6) and tuple(bottom.shape) == (C, 2)
    for c in range(C):
        torch.testing.assert_close(top[c], prec[:4], atol=0.2, rtol=0.2)
        torch.testing.assert_close(bottom[c], prec.diagonal()[4:], atol=0.2, rtol=0.2)
    # the structure: the tail block of the mass matrix is diagonal
    M = kernel.mass_matrix_adapter.mass_matrix_dense
    sl = kernel._layout.slices
    z0, z1 = sl["z"]
    assert float(M[:, z0, z1 - 1].abs().max()) == 0.0
