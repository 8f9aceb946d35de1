"""NUTS on a plated Gaussian mixture with the assignments enumerated (the reference's tests/infer/mcmc/test_nuts.py:
gmm shape, at size), vectorised chains: leapfrog steps/s with the likelihood through the mixture leaf kernel
(csrc/mixture.hip: the chains are its batch of parameter sets) against the materialised [K, C, N] route
(developer tool).     python tools/bench_nuts_gmm.py [N] [K] [C]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import pyro_amd as pyro
import pyro_amd.distributions as dist
import pyro_amd.ops.contract as contract
from pyro_amd.infer.mcmc import MCMC, NUTS
from pyro_amd.ops.indexing import Vindex

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
C = int(sys.argv[3]) if len(sys.argv) > 3 else 64
g = torch.Generator().manual_seed(0)
data = (torch.randn(N, generator=g) + 4.0 * torch.randint(0, K, (N,), generator=g).float()).to(dev)


def model(data):
    phi = pyro.sample("phi", dist.Dirichlet(torch.ones(K, device=dev)))
    with pyro.plate("num_clusters", K):
        means = pyro.sample("cluster_means", dist.Normal(4.0 * torch.arange(K, device=dev, dtype=torch.float32), 1.0))
    with pyro.plate("data", data.shape[0]):
        a = pyro.sample("assignments", dist.Categorical(phi))
        pyro.sample("obs", dist.Normal(Vindex(means.unsqueeze(-2))[..., a], 1.0), obs=data)


for it, leaf in enumerate((True, True, False, True, False)):
    # (the first run of a process pays its one-time costs -- code objects, captures: discarded)
    contract.FUSED_MIXTURE = leaf
    pyro.set_rng_seed(0)
    kernel = NUTS(model, max_tree_depth=5, max_plate_nesting=1)
    mcmc = MCMC(kernel, num_samples=30, warmup_steps=30, num_chains=C)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mcmc.run(data)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = kernel.num_leapfrog_steps
    m = mcmc.get_samples()["cluster_means"].mean(0)
    if it == 0:
        continue
    print("mixture leaf %-5s N=%d K=%d chains=%d: %.0f leapfrog steps/s (%d leapfrogs, %.2f s); posterior means %s" % (
        leaf, N, K, C, n / dt, n, dt, [round(float(v), 2) for v in m]), flush=True)
contract.FUSED_MIXTURE = True
