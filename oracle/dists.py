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


LOG_PROB = {
    0: lambda v, a, b: normal_log_prob(v, a, b),
    1: lambda v, a, b: bernoulli_logits_log_prob(v, a),
    2: lambda v, a, b: half_cauchy_log_prob(v, a),
    3: lambda v, a, b: log_normal_log_prob(v, a, b),
    4: lambda v, a, b: exponential_log_prob(v, a),
    5: lambda v, a, b: half_normal_log_prob(v, a),
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
