"""Chain diagnostics on the device (reference: pyro/ops/stats.py:13-219).

Same estimators (Gelman-Rubin R-hat, split R-hat, FFT autocorrelation, Geyer's initial
monotone sequence ESS with Stan's multi-chain rho) computed as batched tensor ops over
``[chain, sample, ...]`` so 1024 chains x 100 dims are one FFT batch on the GPU."""
import numbers

import torch


def _to_sample_chain_first(x, chain_dim, sample_dim):
    assert x.dim() >= 2
    chain_dim = chain_dim % x.dim()
    sample_dim = sample_dim % x.dim()
    assert chain_dim != sample_dim
    rest = [d for d in range(x.dim()) if d not in (chain_dim, sample_dim)]
    return x.permute([sample_dim, chain_dim] + rest)   # [N, C, ...]


def _chain_variance_stats(x):
    """x: [N, C, ...] -> (var_within, var_estimator) (reference: stats.py:13-29)."""
    N, C = x.size(0), x.size(1)
    chain_var = x.var(dim=0)
    var_within = chain_var.mean(dim=0)
    var_estimator = (N - 1) / N * var_within
    if C > 1:
        var_estimator = var_estimator + x.mean(dim=0).var(dim=0)
    else:
        var_within = var_estimator
    return var_within, var_estimator


def gelman_rubin(input, chain_dim=0, sample_dim=1):
    assert input.size(sample_dim) >= 2 and input.size(chain_dim) >= 2
    x = _to_sample_chain_first(input, chain_dim, sample_dim)
    var_within, var_estimator = _chain_variance_stats(x)
    return (var_estimator / var_within).sqrt()


def split_gelman_rubin(input, chain_dim=0, sample_dim=1):
    assert input.size(sample_dim) >= 4
    x = _to_sample_chain_first(input, chain_dim, sample_dim)
    half = x.size(0) // 2
    x = torch.cat([x[:half], x[-half:]], dim=1)     # [half, 2C, ...]
    var_within, var_estimator = _chain_variance_stats(x)
    return (var_estimator / var_within).sqrt()


def _next_fast_len(n):
    """Smallest 2^a 3^b 5^c >= n."""
    if n <= 2:
        return 2
    while True:
        m = n
        for p in (2, 3, 5):
            while m % p == 0:
                m //= p
        if m == 1:
            return n
        n += 1


def autocorrelation(input, dim=0):
    """FFT autocorrelation along ``dim`` (reference: stats.py:87-128)."""
    N = input.size(dim)
    M2 = 2 * _next_fast_len(N)
    x = input.transpose(dim, -1)
    centered = x - x.mean(dim=-1, keepdim=True)
    f = torch.fft.rfft(centered, n=M2)
    gram = f.real.pow(2) + f.imag.pow(2)
    ac = torch.fft.irfft(gram, n=M2)[..., :N]
    ac = ac / torch.arange(N, 0, -1, dtype=input.dtype, device=input.device)
    variance = ac[..., :1]
    constant = (variance == 0).expand_as(ac)
    ac = ac / variance.clamp(min=torch.finfo(variance.dtype).tiny)
    ac = torch.where(constant, torch.ones_like(ac), ac)
    return ac.transpose(dim, -1)


def autocovariance(input, dim=0):
    return autocorrelation(input, dim) * input.var(dim, unbiased=False, keepdim=True)


def effective_sample_size(input, chain_dim=0, sample_dim=1):
    """ESS = chains * draws / tau with tau = 2 sum_t P_t - 1, where P_t = rho_{2t} + rho_{2t+1} are the pair
    sums of the multi-chain autocorrelation and the sum runs over Geyer's initial monotone sequence: a negative
    pair sum is noise (counted as 0), and no pair sum may exceed the one before it.  Same estimator as the
    reference's (pyro/ops/stats.py:162-219, pinned there against arviz: 52.64 for arange(1000) in 100 chains)."""
    draws_first = _to_sample_chain_first(input, chain_dim, sample_dim)
    n, chains = draws_first.shape[:2]
    assert n >= 2
    within, pooled = _chain_variance_stats(draws_first)
    # rho_t = 1 - (W - mean_c acov_t) / var+ ; lag 0 is 1 by definition
    rho = 1 - (within - autocovariance(draws_first, dim=0).mean(dim=1)) / pooled
    rho[0] = 1
    pairs = rho[:n - n % 2].unflatten(0, (n // 2, 2)).sum(dim=1)
    tail = pairs[1:].clamp(min=0)
    if tail.size(0) > 0:
        tail = torch.cummin(tail, dim=0).values
    tau = 2 * (pairs[0] + tail.sum(dim=0)) - 1
    return chains * n / tau


def quantile(input, probs, dim=0):
    """Linear-interpolation quantiles (reference: stats.py:236-262)."""
    scalar = isinstance(probs, numbers.Number)
    probs = torch.as_tensor(probs, dtype=input.dtype, device=input.device).reshape(-1)
    sorted_input = input.sort(dim)[0]
    max_index = input.size(dim) - 1
    idx = probs * max_index
    below = idx.long()
    above = (below + 1).clamp(max=max_index)
    qa = sorted_input.index_select(dim, above)
    qb = sorted_input.index_select(dim, below)
    shape = [1] * input.dim()
    shape[dim] = idx.numel()
    wa = (idx - below.type_as(idx)).reshape(shape)
    q = (1 - wa) * qb + wa * qa
    return q.squeeze(dim) if scalar else q


def pi(input, prob, dim=0):
    return quantile(input, [(1 - prob) / 2, (1 + prob) / 2], dim)


def hpdi(input, prob, dim=0):
    """Highest posterior density interval (reference: stats.py:346-369)."""
    sorted_input = input.sort(dim)[0]
    mass = input.size(dim)
    index_length = int(prob * mass)
    intervals_left = sorted_input.narrow(dim, 0, mass - index_length)
    intervals_right = sorted_input.narrow(dim, index_length, mass - index_length)
    index_start = (intervals_right - intervals_left).argmin(dim, keepdim=True)
    lo = intervals_left.gather(dim, index_start)
    hi = intervals_right.gather(dim, index_start)
    return torch.cat([lo, hi], dim)


def _cummin(input):
    """Cumulative minimum along dim 0 (the reference builds an N x N mask for it, stats.py:142-159)."""
    return torch.cummin(input, dim=0)[0]


def resample(input, num_samples, dim=0, replacement=False):
    """``num_samples`` entries of ``input`` along ``dim``, drawn uniformly (stats.py:222-233)."""
    weights = torch.ones(input.size(dim), dtype=input.dtype, device=input.device)
    indices = torch.multinomial(weights, num_samples, replacement)
    return input.index_select(dim, indices)
