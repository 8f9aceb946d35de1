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
                              max_tree_depth=6, dtype=torch.float64):
    """NUTS kernel (no adaptation) driven through MCMC, chain-for-chain against the recursive
    restatement of the reference with the same keyed draws."""
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
                     kernel.step_size.clone(), kernel.inverse_mass_matrix.clone(),
                     kernel.num_leapfrog_steps, mcmc.diagnostics()))
    a, b = outs
    assert a[3] == b[3], (a[3], b[3])                       # identical trees
    torch.testing.assert_close(a[1], b[1], rtol=rtol, atol=0)
    torch.testing.assert_close(a[2], b[2], rtol=rtol, atol=0)
    torch.testing.assert_close(a[0], b[0], rtol=rtol, atol=rtol)
    torch.testing.assert_close(a[4]["acceptance rate"], b[4]["acceptance rate"])
    assert a[4]["divergences"] == b[4]["divergences"]
    assert abs(a[4]["mean tree depth"] - b[4]["mean tree depth"]) < 1e-12
