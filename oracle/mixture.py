"""TEST INFRASTRUCTURE -- the leaf of a plated mixture under TraceEnum_ELBO, restated in numpy (float64).

Reference: pyro/infer/traceenum_elbo.py:112-214 (_compute_model_factors: the observed site's log_prob against every
value of the enumerated assignment, a [K, N] tensor from pyro/poutine/trace_struct.py:248-288, beside the
assignment's own log-probabilities) contracted by pyro/ops/contract.py:79-160 (_contract_component: logsumexp over
the enumerated dim through pyro/ops/einsum/torch_log.py:12-49, then the plate product = a sum over n):

    S = sum_n log sum_k exp(a[k] + log p(x[n] | p0[k], p1[k]))
    dS/da[k] = sum_n r[n, k],   dS/dp0[k] = sum_n r[n, k] d log p / d p0,   dS/dp1[k] likewise,
    r[n, k] = softmax_k(a[k] + log p(x[n] | .)) -- the posterior responsibilities autograd arrives at.

This is the arithmetic of pyro_amd/csrc/mixture.hip (pa_mixture_fwd_bwd).  Pinned by
tests/test_oracle_vs_golden.py::test_mixture_oracle_against_the_reference_operators on torch.distributions' log_prob
+ torch.logsumexp + autograd in float64 -- the operators the reference's path executes -- and end to end, through the
kernel, by tests/test_enum_gpu.py::test_gmm_matches_reference on tests/golden/enum.npz (the unmodified reference's
TraceEnum_ELBO loss and gradients of a plated Gaussian mixture, whole plate and subsampled).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline legs may import this module.
"""
import numpy as np

from . import dists


def mixture_fwd_bwd(dist_id, x, a, p0, p1=None):
    """x [N]; a [K]; p0, p1: [K] or scalars.  -> (S, dS/da [K], dS/dp0 [K], dS/dp1 [K]): the parameter gradients
    PER COMPONENT (a parameter shared by all components takes their sum)."""
    x = np.asarray(x, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    K = a.shape[0]
    p0 = np.broadcast_to(np.asarray(p0, dtype=np.float64), (K,))
    p1v = np.zeros(K) if p1 is None else np.broadcast_to(np.asarray(p1, dtype=np.float64), (K,))
    S = 0.0
    da, d0, d1 = np.zeros(K), np.zeros(K), np.zeros(K)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = a[None, :] + dists.LOG_PROB[dist_id](x[:, None], p0[None, :], p1v[None, :])     # [N, K]
        m = t.max(axis=1, keepdims=True)
        dead = ~np.isfinite(m[:, 0]) & (m[:, 0] < 0)
        e = np.where(dead[:, None], 0.0, np.exp(t - np.where(dead[:, None], 0.0, m)))
        s = e.sum(axis=1, keepdims=True)
        lse = np.where(dead, -np.inf, m[:, 0] + np.log(np.where(dead[:, None], 1.0, s))[:, 0])
        r = np.where(dead[:, None], 0.0, e / np.where(dead[:, None], 1.0, s))
        _, g0, g1 = dists.log_prob_grad(dist_id, x[:, None], p0[None, :], p1v[None, :])
        g0 = np.broadcast_to(g0, t.shape)
        g1 = np.broadcast_to(g1, t.shape)
        S = lse.sum()
        da = r.sum(axis=0)
        d0 = np.where(r > 0, r * g0, 0.0).sum(axis=0)
        d1 = np.where(r > 0, r * g1, 0.0).sum(axis=0)
    return S, da, d0, d1


def mixture_diag_normal_fwd_bwd(x, a, loc, scale):
    """The same leaf for event-shaped observations (Normal(loc[z], scale).to_event(1): Independent.log_prob sums the
    features, torch/distributions/independent.py:96-98, INSIDE the logsumexp): x [N, D]; a [K]; loc, scale [K, D].
    -> (S, dS/da [K], dS/dloc [K, D], dS/dscale [K, D])."""
    x = np.asarray(x, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    K = a.shape[0]
    D = x.shape[1]
    loc = np.broadcast_to(np.asarray(loc, dtype=np.float64), (K, D))
    scale = np.broadcast_to(np.asarray(scale, dtype=np.float64), (K, D))
    z = (x[:, None, :] - loc[None]) / scale[None]                                     # [N, K, D]
    with np.errstate(divide="ignore", invalid="ignore"):
        t = a[None, :] + (-0.5 * z * z - np.log(scale)[None] - 0.5 * np.log(2 * np.pi)).sum(-1)
        m = t.max(axis=1, keepdims=True)
        dead = ~np.isfinite(m[:, 0])
        e = np.where(dead[:, None], 0.0, np.exp(t - np.where(dead[:, None], 0.0, m)))
        ssum = e.sum(axis=1, keepdims=True)
        lse = np.where(dead, -np.inf, m[:, 0] + np.log(np.where(dead[:, None], 1.0, ssum))[:, 0])
        r = np.where(dead[:, None], 0.0, e / np.where(dead[:, None], 1.0, ssum))       # [N, K]
    dl = (r[:, :, None] * z / scale[None]).sum(0)
    dc = (r[:, :, None] * (z * z - 1.0) / scale[None]).sum(0)
    return lse.sum(), r.sum(0), dl, dc
