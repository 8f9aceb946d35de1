"""NUTS on a MODEL (Bayesian logistic regression through the handlers, not a closed-form
potential): leapfrog steps/s with the potential evaluated eagerly vs replayed as a hipGraph
(jit_compile=True).  Developer tool."""
import sys
import time

import torch

sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd.infer.mcmc import MCMC, NUTS
from tests import mcmc_cases as mc

dev = torch.device("cuda:0")
N, D, C = 100_000, 32, 256
g = torch.Generator().manual_seed(0)
Xc = torch.randn((N, D), generator=g)
yc = (torch.rand((N,), generator=g) < torch.sigmoid(Xc @ torch.randn(D, generator=g) * 0.3)).float()
X, y = Xc.to(dev), yc.to(dev)
for jit in (False, True):
    pyro.set_rng_seed(1)
    kernel = NUTS(mc.logreg_mcmc_model, max_tree_depth=6, jit_compile=jit)
    mcmc = MCMC(kernel, num_samples=20, warmup_steps=30, num_chains=C)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mcmc.run(X, y)
    n = kernel.num_leapfrog_steps
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("jit_compile=%s: %.2f s, %d leapfrogs (%d chains), %.0f leapfrog/s, %.0f potential evaluations/s"
          % (jit, dt, n, C, n / dt, n / C / dt))
