"""TEST INFRASTRUCTURE -- numpy restatement of the plated Bernoulli-logits GLM site and of the
Trace_ELBO surrogate for the BASELINE config-2 model (Bayesian logistic regression).

Reference chain of operations restated here (P vectorised particles, plate size N):
  logits = w @ X^T + b                     user model
  log p = Bernoulli(logits).log_prob(y)    torch: torch/distributions/bernoulli.py:121-125
  scale_and_mask, .sum()                   pyro/poutine/trace_struct.py:264-278,
                                           pyro/distributions/util.py:311-328
  backward                                 pyro/infer/trace_elbo.py:153-157
"""
import numpy as np

from .dists import _sigmoid, _softplus


def glm_bernoulli_fwd_bwd(X, y, w, b=None, mask=None, scale=1.0):
    """Returns ll[P], gw[P,D], gb[P] in float64."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    logits = w @ X.T
    if b is not None:
        logits = logits + np.asarray(b, dtype=np.float64)[:, None]
    lp = y[None, :] * logits - _softplus(logits)
    g = y[None, :] - _sigmoid(logits)
    if mask is not None:
        m = np.asarray(mask, dtype=bool)[None, :]
        lp = np.where(m, lp, 0.0)
        g = np.where(m, g, 0.0)
    return scale * lp.sum(1), scale * (g @ X), scale * g.sum(1)


def glm_bernoulli_grouped_fwd_bwd(X, y, w, group_of_row, b=None, mask=None, scale=1.0):
    """Hierarchical variant: logit[p, n] = w[p, g(n), :] . x_n + b[p].
    w: [P, G, D].  Returns ll[P], gw[P, G, D], gb[P] in float64."""
    X = np.asarray(X, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    g_of = np.asarray(group_of_row)
    logits = np.einsum("pnd,nd->pn", w[:, g_of, :], X)
    if b is not None:
        logits = logits + np.asarray(b, dtype=np.float64)[:, None]
    lp = y[None, :] * logits - _softplus(logits)
    g = y[None, :] - _sigmoid(logits)
    if mask is not None:
        m = np.asarray(mask, dtype=bool)[None, :]
        lp = np.where(m, lp, 0.0)
        g = np.where(m, g, 0.0)
    gw = np.zeros_like(w)
    for grp in range(w.shape[1]):
        sel = g_of == grp
        gw[:, grp, :] = g[:, sel] @ X[sel]
    return scale * lp.sum(1), scale * gw, scale * g.sum(1)
