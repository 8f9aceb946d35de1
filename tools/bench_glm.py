"""GLM kernel micro-benchmark + correctness spot check (developer tool)."""
import sys

import torch

sys.path.insert(0, ".")
from pyro_amd import kernels as k
from tools.bench_kernels import timeit

dev = torch.device("cuda:0")
QUICK = "--quick" in sys.argv
CONFIGS = [(1_000_000, 32, 64), (1_000_000, 32, 32), (1_000_000, 64, 32), (10_000_000, 32, 64),
           (1_000_000, 8, 64), (1_000_000, 128, 64), (1_000_000, 64, 64), (1_000_000, 32, 128),
           (1_000_000, 32, 8), (1_000_000, 32, 1), (1_000_000, 20, 64)]   # sweep D in {8, 128}
if QUICK:
    CONFIGS = CONFIGS[:1]
for variant in (k.GLM_AUTO, k.GLM_EXACT_F32):
  k.glm_set_variant(variant)
  print("variant", "auto (rows / bf16x3)" if variant == k.GLM_AUTO else "exact f32")
  for (N, D, P) in CONFIGS:
      X = torch.randn((N, D), device=dev)
      y = (torch.rand((N,), device=dev) < 0.5).float()
      w = torch.randn((P, D), device=dev) * 0.2
      b = torch.randn((P,), device=dev)
      us = timeit(lambda: k.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0), n=8 if QUICK else 30, warm=2 if QUICK else 5)
      ll, gw, gb = k.glm_bernoulli_fwd_bwd(X, y, w, b, None, 1.0)
      if N <= 1_000_000:
          Xd, wd = X.double(), w.double()
          l = wd @ Xd.t() + b.double()[:, None]
          llr = (y.double() * l - torch.nn.functional.softplus(l)).sum(1)
          g = y.double() - torch.sigmoid(l)
          err = ((ll.double() - llr).abs() / llr.abs()).max().item()
          errg = ((gw.double() - g @ Xd).abs().max() / (g @ Xd).abs().max()).item()
      else:
          err = errg = float("nan")
      print(f"glm N={N} D={D} P={P}: {us:8.1f} us {N*(4*D+4)/us/1e6:6.3f} TB/s {4.0*P*N*D/us/1e6:6.2f} TFLOP/s  rel err ll {err:.1e} gw {errg:.1e}")
      del X, y
