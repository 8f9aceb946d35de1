"""TEST INFRASTRUCTURE -- numpy restatement of the per-site arithmetic of the ELBO hot path.

The reference's distribution classes are third-party torch.distributions classes with a Pyro
mixin (pyro/distributions/torch.py:395-408); the formulas below restate the installed torch
(2.10.0) sources cited per function.  Gradients are the analytic derivatives autograd
produces for those formulas.
"""
import math

import numpy as np

HALF_LOG_2PI = 0.5 * math.log(2 * math.pi)


def _softplus(x):
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def _sigmoid(x):
    e = np.exp(-np.abs(x))
    return np.where(x >= 0, 1 / (1 + e), e / (1 + e))


# --- log_prob ---------------------------------------------------------------------------------

def normal_log_prob(v, loc, scale):
    """torch: torch/distributions/normal.py:88-103."""
    return -((v - loc) ** 2) / (2 * scale ** 2) - np.log(scale) - HALF_LOG_2PI


def bernoulli_logits_log_prob(v, logits):
    """torch: torch/distributions/bernoulli.py:121-125 (= -binary_cross_entropy_with_logits)."""
    return v * logits - _softplus(logits)


def half_cauchy_log_prob(v, scale):
    """torch: torch/distributions/half_cauchy.py:74-83 (Cauchy.log_prob + log 2, -inf for v<0)."""
    lp = math.log(2) - math.log(math.pi) - np.log(scale) - np.log1p((v / scale) ** 2)
    return np.where(v >= 0, lp, -np.inf)


def log_normal_log_prob(v, loc, scale):
    """torch: TransformedDistribution(Normal, ExpTransform).log_prob
    (torch/distributions/transformed_distribution.py:157-180)."""
    lv = np.log(v)
    return normal_log_prob(lv, loc, scale) - lv


def exponential_log_prob(v, rate):
    """torch: torch/distributions/exponential.py (rate.log() - rate * value)."""
    return np.log(rate) - rate * v


def half_normal_log_prob(v, scale):
    """torch: torch/distributions/half_normal.py:66-71."""
    lp = normal_log_prob(v, 0.0, scale) + math.log(2)
    return np.where(v >= 0, lp, -np.inf)


def _xlogy(x, y):
    """torch.xlogy: 0 where x == 0 whatever y is."""
    x, y = np.broadcast_arrays(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64))
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(x == 0, 0.0, x * np.log(y))


def gamma_log_prob(v, concentration, rate):
    """torch: torch/distributions/gamma.py log_prob
    (xlogy(concentration, rate) + xlogy(concentration - 1, value) - rate * value - lgamma)."""
    from scipy.special import gammaln
    return _xlogy(concentration, rate) + _xlogy(concentration - 1, v) - rate * v - gammaln(concentration)


def beta_log_prob(v, c1, c0):
    """torch: torch/distributions/beta.py log_prob = Dirichlet([c1, c0]).log_prob([v, 1 - v])
    (torch/distributions/dirichlet.py log_prob: sum xlogy(c - 1, x) + lgamma(sum c) - sum lgamma(c))."""
    from scipy.special import gammaln
    return _xlogy(c1 - 1, v) + _xlogy(c0 - 1, 1 - v) + gammaln(c1 + c0) - gammaln(c1) - gammaln(c0)


def poisson_log_prob(v, rate):
    """torch: torch/distributions/poisson.py log_prob (value.xlogy(rate) - rate - lgamma(value + 1))."""
    from scipy.special import gammaln
    return _xlogy(v, rate) - rate - gammaln(v + 1)


def binomial_logits_log_prob(v, logits, total_count):
    """pyro/distributions/torch.py:83-101 with approx_log_prob_tol = 0:
    k * logits - n * softplus(logits) + log C(n, k)   (pyro/ops/special.py log_binomial)."""
    from scipy.special import gammaln
    n = total_count
    return v * logits - n * _softplus(logits) + gammaln(n + 1) - gammaln(v + 1) - gammaln(n - v + 1)


def dirichlet_log_prob(x, concentration):
    """torch: torch/distributions/dirichlet.py log_prob
    (xlogy(concentration - 1, value).sum(-1) + lgamma(concentration.sum(-1)) - lgamma(concentration).sum(-1))."""
    from scipy.special import gammaln
    c = np.asarray(concentration, dtype=np.float64)
    return _xlogy(c - 1, x).sum(-1) + gammaln(c.sum(-1)) - gammaln(c).sum(-1)


def dirichlet_log_prob_grad(g, x, concentration):
    """(d/dx, d/dconcentration) of sum(g * log_prob), both of the broadcast [..., K] shape."""
    from scipy.special import digamma
    x = np.asarray(x, dtype=np.float64)
    c = np.asarray(concentration, dtype=np.float64)
    shape = np.broadcast(x, c).shape
    x, c = np.broadcast_to(x, shape), np.broadcast_to(c, shape)
    g = np.broadcast_to(np.asarray(g, dtype=np.float64), shape[:-1])[..., None]
    return g * (c - 1) / x, g * (np.log(x) + digamma(c.sum(-1, keepdims=True)) - digamma(c))


def kl_normal_normal(lq, sq, lp, sp):
    """torch: torch/distributions/kl.py _kl_normal_normal."""
    var_ratio = (sq / sp) ** 2
    t1 = ((lq - lp) / sp) ** 2
    return 0.5 * (var_ratio + t1 - 1 - np.log(var_ratio))


def kl_normal_loc_half(lq, lp, sp):
    """-KL(N(lq,sq) || N(lp,sp)) = this + kl_normal_scale_half(sq, sp)   (include/pyro_amd.h)."""
    return -((lq - lp) ** 2) / (2 * sp ** 2) - np.log(sp)


def kl_normal_scale_half(sq, sp):
    return np.log(sq) + 0.5 - sq ** 2 / (2 * sp ** 2)


LOG_PROB = {
    0: lambda v, a, b: normal_log_prob(v, a, b),
    1: lambda v, a, b: bernoulli_logits_log_prob(v, a),
    2: lambda v, a, b: half_cauchy_log_prob(v, a),
    3: lambda v, a, b: log_normal_log_prob(v, a, b),
    4: lambda v, a, b: exponential_log_prob(v, a),
    5: lambda v, a, b: half_normal_log_prob(v, a),
    6: lambda v, a, b: gamma_log_prob(v, a, b),
    7: lambda v, a, b: beta_log_prob(v, a, b),
    8: lambda v, a, b: poisson_log_prob(v, a),
    9: lambda v, a, b: binomial_logits_log_prob(v, a, b),
    10: lambda v, a, b: kl_normal_loc_half(v, a, b),
    11: lambda v, a, b: kl_normal_scale_half(v, a),
}


# --- gradients (d log_prob / d value, d p0, d p1) ---------------------------------------------

def normal_grad(v, a, b):
    d = v - a
    da = d / b ** 2
    return -da, da, d * d / b ** 3 - 1 / b


def log_prob_grad(dist_id, v, a, b):
    v, a = np.asarray(v), np.asarray(a)
    z = np.zeros(np.broadcast(v, a).shape)
    if dist_id == 0:
        return normal_grad(v, a, b)
    if dist_id == 1:
        return a + z, v - _sigmoid(a), z
    if dist_id == 2:
        den = a * a + v * v
        return -2 * v / den, (v * v - a * a) / (a * den), z
    if dist_id == 3:
        lv = np.log(v)
        dn, da, db = normal_grad(lv, a, b)
        return (dn - 1) / v, da, db
    if dist_id == 4:
        return -a + z, 1 / a - v, z
    if dist_id == 5:
        dv, _, db = normal_grad(v, 0.0, a)
        return dv, db, z
    if dist_id == 10:
        return normal_grad(v, a, b)
    if dist_id == 11:
        return 1 / v - v / a ** 2, v ** 2 / a ** 3, z
    if dist_id in (6, 7, 8, 9):
        from scipy.special import digamma
        if dist_id == 6:
            return (a - 1) / v - b, np.log(b) + np.log(v) - digamma(a) + z, a / b - v
        if dist_id == 7:
            pab = digamma(a + b)
            return ((a - 1) / v - (b - 1) / (1 - v), np.log(v) + pab - digamma(a),
                    np.log(1 - v) + pab - digamma(b))
        if dist_id == 8:
            return np.log(a) - digamma(v + 1), v / a - 1, z
        pnk = digamma(b - v + 1)
        return (a - digamma(v + 1) + pnk, v - b * _sigmoid(a),
                digamma(b + 1) - pnk - _softplus(a))
    raise ValueError(dist_id)


# --- scale_and_mask + plate sum -----------------------------------------------------------------

def scale_and_mask(x, scale=1.0, mask=None):
    """pyro/distributions/util.py:311-328."""
    if mask is None:
        return x * scale
    return np.where(mask, x * scale, 0.0)


def log_prob_sum(dist_id, v, a, b, mask=None, scale=1.0):
    """trace_struct.py:264-278: log_prob -> scale_and_mask -> sum over the last axis (per row)."""
    lp = LOG_PROB[dist_id](v, a, b)
    shape = np.broadcast(v, a, b if b is not None else 0.0).shape
    lp = np.broadcast_to(lp, shape)
    return scale_and_mask(lp, scale, mask).sum(-1)


def log_prob_sum_grad(dist_id, g_row, v, a, b, mask=None, scale=1.0):
    """Full-size gradients of sum_c g_row[r] * scale_and_mask(log_prob)[r,c]."""
    dv, da, db = log_prob_grad(dist_id, v, a, b)
    w = np.asarray(g_row)[..., None] * scale
    outs = []
    for d in (dv, da, db):
        x = w * d
        if mask is not None:
            x = np.where(mask, x, 0.0)
        outs.append(x)
    return outs
