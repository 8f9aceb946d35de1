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


for leaf, fused in ((True, True), (False, True), (False, False)):
    # leaf: the mixture leaf kernel (csrc/mixture.hip: no [K, N] tensor at all); fused: the elimination through
    # pa_logsumexp_terms over a materialised [K, N] likelihood (round 3); neither: adds + torch.logsumexp
    contract.FUSED_MIXTURE, contract.FUSED_SUMPRODUCT = leaf, fused
    for graph in (False, True):
        pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
        svi = SVI(model, guide, pyro.optim.Adam({"lr": 0.01}), TraceEnum_ELBO(max_plate_nesting=1),
                  hip_graph=graph, graph_warmup=3)
        losses = [svi.step(data) for _ in range(8)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            l = svi.step(data)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        print("mixture_leaf", leaf, "fused_sumproduct", fused, "graph", graph, "ms/step %.3f" % (dt * 1e3),
              "loss %.1f" % losses[-1], flush=True)
        svi.release()
contract.FUSED_MIXTURE = contract.FUSED_SUMPRODUCT = True
# the leaf kernel alone
from pyro_amd import _lib, kernels
a = torch.log(torch.full((K,), 1.0 / K, device=dev))
p0 = 3.0 * torch.arange(K, device=dev, dtype=torch.float32)
p1 = torch.ones(1, device=dev)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
kernels.mixture_fwd_bwd(_lib.DIST_NORMAL, data, a, p0, 1, p1, 0)
s.record()
for _ in range(20):
    kernels.mixture_fwd_bwd(_lib.DIST_NORMAL, data, a, p0, 1, p1, 0)
e.record(); torch.cuda.synchronize()
print("pa_mixture_fwd_bwd alone (kernel + finalize + workspace): %.1f us for N=%d K=%d" % (s.elapsed_time(e) * 50, N, K))

# ---- the multi-dimensional mixture: Normal(loc[z], 1).to_event(1) over D features ------------------------------
D = 4
datad = (torch.randn(N, D, generator=g) + 3 * torch.randn(K, D, generator=g)[torch.randint(0, K, (N,), generator=g)]).to(dev)


@config_enumerate
def modeld(data):
    w = pyro.sample("w", dist.Dirichlet(torch.ones(K, device=dev)))
    with pyro.plate("k", K):
        loc = pyro.sample("loc", dist.Normal(torch.zeros(D, device=dev), 20.0).to_event(1))
    with pyro.plate("n", N):
        z = pyro.sample("z", dist.Categorical(w))
        pyro.sample("x", dist.Normal(loc[z], 1.0).to_event(1), obs=data)


lq0 = (3.0 * torch.randn(K, D, generator=g)).to(dev)


def guided(data):
    wq = pyro.param("wq", torch.ones(K, device=dev), constraint=constraints.positive)
    lq = pyro.param("lqd", lq0)
    pyro.sample("w", dist.Dirichlet(wq))
    with pyro.plate("k", K):
        pyro.sample("loc", dist.Normal(lq, 0.5).to_event(1))


for leaf in (True, False):
    contract.FUSED_MIXTURE = leaf
    pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
    svi = SVI(modeld, guided, pyro.optim.Adam({"lr": 0.01}), TraceEnum_ELBO(max_plate_nesting=1), hip_graph=True,
              graph_warmup=3)
    losses = [svi.step(datad) for _ in range(8)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        svi.step(datad)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print("D=%d mixture_leaf" % D, leaf, "graph True ms/step %.3f" % (dt * 1e3), "loss %.1f" % losses[-1], flush=True)
    svi.release()
contract.FUSED_MIXTURE = True
