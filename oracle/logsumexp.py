"""TEST INFRASTRUCTURE -- one elimination step of the plated sum-product in log space, restated in
numpy after the reference's chain of operations (pyro/ops/contract.py:79-160 sums the aligned
log-factors that mention the variable; pyro/ops/einsum/torch_log.py:14-55 takes the max-shifted
log-sum-exp over it) and its gradient (the posterior weights of the eliminated variable).
"""
import numpy as np


def logsumexp_terms(terms, frame, rdim):
    """terms: arrays broadcastable to ``frame``.  Returns (out [frame without rdim], total [frame])."""
    total = np.zeros(frame, dtype=np.float64)
    for t in terms:
        total = total + np.broadcast_to(np.asarray(t, dtype=np.float64), frame)
    shift = total.max(rdim, keepdims=True) if total.shape[rdim] else np.full(
        frame[:rdim] + (1,) + frame[rdim + 1:], -np.inf)
    shift = np.where(np.isfinite(shift), shift, 0.0)      # torch_log.py: clamp(min=finfo.min)
    with np.errstate(divide="ignore"):
        out = np.log(np.exp(total - shift).sum(rdim, keepdims=True)) + shift
    return np.squeeze(out, rdim), total


def logsumexp_terms_grad(terms, frame, rdim, g_out):
    """G[frame] = g_out[kept] * softmax over rdim of the summed terms (0 where the column is -inf)."""
    out, total = logsumexp_terms(terms, frame, rdim)
    o = np.expand_dims(out, rdim)
    with np.errstate(invalid="ignore"):
        w = np.where(np.isfinite(o) & np.isfinite(total), np.exp(total - o), 0.0)
    return np.expand_dims(np.asarray(g_out, dtype=np.float64), rdim) * w
