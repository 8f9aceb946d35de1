"""A plated Gaussian mixture with the assignment enumerated (TraceEnum_ELBO): step time with the
eliminations through pa_logsumexp_terms against the torch route (developer tool)."""
import sys
import time

import torch

sys.path.insert(0, ".")
import pyro_amd as pyro
import pyro_amd.distributions as dist
from pyro_amd.infer import SVI, TraceEnum_ELBO, config_enumerate
from pyro_amd.ops import contract
from torch.distributions import constraints

dev = torch.device("cuda:0")
N, K = 1_000_000, 16
g = torch.Generator().manual_seed(0)
data = (torch.randn(N, generator=g) + 3 * torch.randint(0, K, (N,), generator=g).float()).to(dev)


@config_enumerate
def model(data):
    w = pyro.sample("w", dist.Dirichlet(torch.ones(K, device=dev)))
    with pyro.plate("k", K):
        loc = pyro.sample("loc", dist.Normal(torch.zeros((), device=dev), 20.0))
    with pyro.plate("n", N):
        z = pyro.sample("z", dist.Categorical(w))
        pyro.sample("x", dist.Normal(loc[z], 1.0), obs=data)


def guide(data):
    wq = pyro.param("wq", torch.ones(K, device=dev), constraint=constraints.positive)
    lq = pyro.param("lq", 3.0 * torch.arange(K, device=dev, dtype=torch.float32))
    pyro.sample("w", dist.Dirichlet(wq))
    with pyro.plate("k", K):
        pyro.sample("loc", dist.Normal(lq, 0.5))


for fused in (True, False):
    contract.FUSED_SUMPRODUCT = fused
    for graph in (False, True):
        pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
        svi = SVI(model, guide, pyro.optim.Adam({"lr": 0.01}), TraceEnum_ELBO(max_plate_nesting=1),
                  hip_graph=graph, graph_warmup=3)
        losses = [svi.step(data) for _ in range(8)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            l = svi.step(data)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        print("fused_sumproduct", fused, "graph", graph, "ms/step %.3f" % (dt * 1e3), "loss %.1f" % losses[-1])
contract.FUSED_SUMPRODUCT = True
