"""TEST INFRASTRUCTURE -- the potential of a FLAT model and its gradient, restated in numpy.

Reference: pyro/infer/mcmc/util.py:264-286 (_PEMaker._potential_fn: the model conditioned on the
constrained values of the unconstrained point, U = -(trace.log_prob_sum() + sum of the transforms'
log|det J|)), built by initialize_model (:370-482) with transforms = biject_to(support).inv (:431-447);
pyro/ops/integrator.py:68-94 differentiates it by autograd.  For a model whose latent sites are
scored at parameters that do not depend on other latents and whose one observed site is the
Bernoulli-logits GLM over a latent weight vector / bias, that is

    U(u)   = -( ll_glm(w, b) + sum_sites sum_j [ log p_s(v_j; p0, p1) + log |dv_j / du_j| ] ),  v = T_s(u)
    dU/du  = -( (d log p_s / dv + d ll_glm / dv) dv/du + d log|dv/du| / du )

with T_s the identity (real support) or v = lower + exp(u) (positive / greater-than support:
torch.distributions.constraint_registry biject_to -> ExpTransform, AffineTransform).  This is the
arithmetic of pyro_amd/csrc/nuts_tree.hip::direct_potential (pa_nuts_tree_run_advance_direct,
pa_nuts_direct_potential).  Pinned by tests/test_oracle_vs_golden.py against
tests/golden/mcmc_direct_potential.npz (the unmodified reference's potential_fn + autograd, float64).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs may import this module.
"""
import numpy as np

from . import dists, glm


def direct_potential(sites, z, X=None, y=None, w_site=None, b_site=None):
    """sites: list of dicts {name, dist (oracle.dists id), transform (0 identity / 1 lower + exp), lower,
    p0, p1} in the flat layout's order; z: {name: unconstrained value}.  -> (U, {name: dU/du}).
    A HIERARCHICAL prior: ``par0`` / ``par1`` = the name of the latent site whose constrained value is the
    parameter (w ~ Normal(mu, tau)); flat element j of the site takes element j % len(parent) of the parent
    (a parent broadcast over leading plate dims) and d log p / d parameter flows back to it."""
    v, dvdu, lp, dlp = {}, {}, 0.0, {}
    for s in sites:                                           # every site's constrained value first
        u = np.asarray(z[s["name"]], dtype=np.float64)
        if s["transform"] == 1:
            e = np.exp(u)
            v[s["name"]], dvdu[s["name"]] = s["lower"] + e, e
            lp += u.sum()                                     # log |dv/du| = u
        else:
            v[s["name"]], dvdu[s["name"]] = u, np.ones_like(u)
        dlp[s["name"]] = np.zeros_like(u)
    for s in sites:
        val = v[s["name"]]
        params, pars = [], []
        for key, pk in (("p0", "par0"), ("p1", "par1")):
            if s.get(pk) is not None:
                pv = v[s[pk]].reshape(-1)
                params.append(np.resize(pv, val.size).reshape(val.shape))       # element j <- parent[j % len]
                pars.append(s[pk])
            else:
                params.append(0.0 if s.get(key) is None else np.asarray(s[key], dtype=np.float64))
                pars.append(None)
        a, b = params
        lp += np.sum(dists.LOG_PROB[s["dist"]](val, a, b))
        g = dists.log_prob_grad(s["dist"], val, a, b)
        dlp[s["name"]] = dlp[s["name"]] + np.broadcast_to(g[0], val.shape)
        for t, par in enumerate(pars):
            if par is not None:
                gp = np.broadcast_to(g[1 + t], val.shape).reshape(-1, v[par].size).sum(0)
                dlp[par] = dlp[par] + gp.reshape(v[par].shape)
    if X is not None:
        w = v[w_site].reshape(1, -1)
        b = None if b_site is None else np.asarray(v[b_site]).reshape(1)
        ll, gw, gb = glm.glm_bernoulli_fwd_bwd(X, y, w, b)
        lp += float(ll[0])
        dlp[w_site] = dlp[w_site] + gw[0]
        if b_site is not None:
            dlp[b_site] = dlp[b_site] + np.reshape(gb[0], np.shape(dlp[b_site]))
    grad = {}
    for s in sites:
        n = s["name"]
        grad[n] = -(dlp[n] * dvdu[n] + (1.0 if s["transform"] == 1 else 0.0))
    return -lp, grad


# the models of tests/golden/make_golden.py::g_mcmc_direct_potential (parameters as written there)
def golden_models(D):
    normal_w = dict(name="w", dist=0, transform=0, lower=0.0, p0=np.zeros(D), p1=1.0)
    normal_b = dict(name="b", dist=0, transform=0, lower=0.0, p0=0.0, p1=1.0)
    return {
        "logreg": dict(sites=[normal_b, normal_w], w_site="w", b_site="b"),
        "positive_site": dict(sites=[normal_b, dict(name="tau", dist=5, transform=1, lower=0.0, p0=1.0, p1=None),
                                     normal_w], w_site="w", b_site="b"),
        "all_families": dict(sites=[
            dict(name="a_hc", dist=2, transform=1, lower=0.0, p0=np.full(3, 0.7), p1=None),
            dict(name="c_ln", dist=3, transform=1, lower=0.0, p0=np.array([0.2, -0.4]), p1=np.array([0.5, 1.5])),
            dict(name="d_ex", dist=4, transform=1, lower=0.0, p0=1.3, p1=None),
            dict(name="e_hn", dist=5, transform=1, lower=0.0, p0=np.array([0.8, 2.0]), p1=None),
            dict(name="f_ga", dist=6, transform=1, lower=0.0, p0=np.array([2.5, 0.6]), p1=np.array([1.5, 0.9])),
            dict(name="w", dist=0, transform=0, lower=0.0, p0=np.full(D, 0.1), p1=np.full(D, 2.0))],
            w_site="w", b_site=None),
        "hier_scale": dict(sites=[
            dict(name="tau", dist=5, transform=1, lower=0.0, p0=1.0, p1=None),
            dict(name="w", dist=0, transform=0, lower=0.0, p0=np.zeros(D), par1="tau")], w_site="w", b_site=None),
        "hier_loc_scale": dict(sites=[
            normal_b,
            dict(name="mu", dist=0, transform=0, lower=0.0, p0=np.zeros(D), p1=1.0),
            dict(name="tau", dist=5, transform=1, lower=0.0, p0=np.ones(D), p1=None),
            dict(name="theta", dist=0, transform=0, lower=0.0, par0="mu", par1="tau"),
            dict(name="w", dist=0, transform=0, lower=0.0, par0="mu", par1="tau")], w_site="w", b_site="b"),
    }
