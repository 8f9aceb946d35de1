"""Welford online (co)variance (reference: pyro/ops/welford.py:7-52).

``sample`` may be [D] (one chain) or [C, D] (one independent estimator per vectorised chain;
diagonal only -- the dense estimator is per chain [D, D] as in the reference)."""
import torch


class WelfordCovariance:
    def __init__(self, diagonal=True):
        self.diagonal = diagonal
        self.reset()

    def reset(self):
        self._mean = 0.0
        self._m2 = 0.0
        self.n_samples = 0

    def update(self, sample):
        self.n_samples += 1
        delta_pre = sample - self._mean
        self._mean = self._mean + delta_pre / self.n_samples
        delta_post = sample - self._mean
        if self.diagonal:
            self._m2 = self._m2 + delta_pre * delta_post
        elif sample.dim() == 1:
            self._m2 = self._m2 + torch.outer(delta_post, delta_pre)
        else:
            self._m2 = self._m2 + delta_post.unsqueeze(-1) * delta_pre.unsqueeze(-2)

    def get_covariance(self, regularize=True):
        if self.n_samples < 2:
            raise RuntimeError("Insufficient samples to estimate covariance")
        cov = self._m2 / (self.n_samples - 1)
        if regularize:
            # regularisation from Stan
            scaled_cov = (self.n_samples / (self.n_samples + 5.0)) * cov
            shrinkage = 1e-3 * (5.0 / (self.n_samples + 5.0))
            if self.diagonal:
                cov = scaled_cov + shrinkage
            else:
                eye = torch.eye(scaled_cov.size(-1), dtype=scaled_cov.dtype,
                                device=scaled_cov.device)
                cov = scaled_cov + shrinkage * eye
        return cov
