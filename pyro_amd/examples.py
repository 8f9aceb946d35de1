"""Benchmark / example programs written against the drop-in API (BASELINE.json configs)."""
import torch

from . import distributions as dist
from .primitives import plate, sample


def logreg_model(X, y):
    """BASELINE config 2: Bayesian logistic regression, plate over N data points.
    The logits stay lazy (dist.linear_logits) so the observed site runs the fused one-pass
    GLM kernel; replace it by ``(w @ X.T).squeeze(-2) + b`` for the reference formulation."""
    N, D = X.shape
    w = sample("w", dist.Normal(torch.zeros(D, dtype=X.dtype, device=X.device), 1.0).to_event(1))
    b = sample("b", dist.Normal(torch.zeros((), dtype=X.dtype, device=X.device), 1.0))
    with plate("data", N):
        sample("obs", dist.Bernoulli(logits=dist.linear_logits(X, w, b)), obs=y)


def logreg_model_unfused(X, y):
    N, D = X.shape
    w = sample("w", dist.Normal(torch.zeros(D, dtype=X.dtype, device=X.device), 1.0).to_event(1))
    b = sample("b", dist.Normal(torch.zeros((), dtype=X.dtype, device=X.device), 1.0))
    with plate("data", N):
        logits = w @ X.t()
        logits = logits.squeeze(-2) if logits.dim() > 1 else logits
        sample("obs", dist.Bernoulli(logits=logits + b), obs=y)


def synthetic_logreg_data(N, D, device, seed=0, dtype=torch.float32):
    g = torch.Generator(device=device).manual_seed(seed)
    X = torch.randn((N, D), device=device, dtype=dtype, generator=g)
    w_true = torch.randn((D,), device=device, dtype=dtype, generator=g)
    y = (torch.rand((N,), device=device, dtype=dtype, generator=g) < torch.sigmoid(X @ w_true)).to(dtype)
    return X, y


def correlated_gaussian_precision(D, seed=0, dtype=torch.float64):
    """BASELINE config 3: Sigma = A A^T / D + 0.1 I, Lambda = Sigma^-1 (built in float64)."""
    g = torch.Generator().manual_seed(seed)
    A = torch.randn((D, D), dtype=torch.float64, generator=g)
    Sigma = A @ A.T / D + 0.1 * torch.eye(D, dtype=torch.float64)
    Lam = torch.linalg.inv(Sigma)
    return Sigma.to(dtype), (0.5 * (Lam + Lam.T)).to(dtype)
