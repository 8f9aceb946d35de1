"""Streaming mean / scatter statistics of the warm-up draws, from which HMC / NUTS build their
inverse mass matrix (Welford's update; reference: pyro/ops/welford.py:7-52, regularisation as in
Stan's ``welford_*_estimator``).

Written around the record the persistent NUTS kernel keeps per chain (``[C, 2, D] = {mean,
scatter}`` for the diagonal estimator): ``to_record`` / ``from_record`` move it to and from the
device path.  A draw is ``[D]`` (one chain) or ``[C, D]`` (one independent estimator per
vectorised chain); the dense estimator keeps a ``[..., D, D]`` scatter matrix per chain.
"""
import torch

# the estimate is shrunk towards SHRINK_TARGET * I as if SHRINK_COUNT extra draws had been seen
SHRINK_COUNT = 5.0
SHRINK_TARGET = 1e-3


class WelfordCovariance:
    def __init__(self, diagonal=True):
        self.diagonal = diagonal
        self.reset()

    def reset(self):
        self.n_samples = 0
        self.mean = 0.0
        self.scatter = 0.0          # sum of (x - mean_before)(x - mean_after): [.., D] or [.., D, D]

    def update(self, sample):
        self.n_samples += 1
        before = sample - self.mean
        self.mean = self.mean + before / self.n_samples
        after = sample - self.mean
        if self.diagonal:
            self.scatter = self.scatter + before * after
        else:
            # outer product per chain; `after` indexes rows so that the sum stays symmetric up to
            # rounding exactly as the one-chain torch.outer(after, before) form does
            self.scatter = self.scatter + after.unsqueeze(-1) * before.unsqueeze(-2)

    def get_covariance(self, regularize=True):
        n = self.n_samples
        if n < 2:
            raise RuntimeError("Insufficient samples to estimate covariance")
        cov = self.scatter / (n - 1)
        if not regularize:
            return cov
        keep = n / (n + SHRINK_COUNT)
        floor = SHRINK_TARGET * (SHRINK_COUNT / (n + SHRINK_COUNT))
        if self.diagonal:
            return keep * cov + floor
        d = cov.size(-1)
        return keep * cov + floor * torch.eye(d, dtype=cov.dtype, device=cov.device)

    # ---- the kernel's record (diagonal estimator) ----------------------------------------------
    def to_record(self, chains, dim, dtype, device):
        rec = torch.zeros((chains, 2, dim), dtype=dtype, device=device)
        rec[:, 0] = self.mean
        rec[:, 1] = self.scatter
        return rec

    def from_record(self, rec, draws_taken):
        self.mean, self.scatter = rec[:, 0].clone(), rec[:, 1].clone()
        self.n_samples += draws_taken
