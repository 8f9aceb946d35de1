#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, time, torch
sys.path.insert(0, "tools")
import pyro_amd as pyro
import pyro_amd.distributions as dist
from torch.distributions import constraints
from pyro_amd import examples
from pyro_amd.infer import TraceMeanField_ELBO, trace_mean_field_elbo as tmf, SVI
dev = torch.device("cuda:0")
N, D = 1_000_000, 32
X, y = examples.synthetic_logreg_data(N, D, dev, seed=0)
def guide(X, y):
    wl = pyro.param("wl", torch.zeros(D, device=dev)); ws = pyro.param("ws", 0.1 * torch.ones(D, device=dev), constraint=constraints.positive)
    bl = pyro.param("bl", torch.zeros((), device=dev)); bs = pyro.param("bs", 0.1 * torch.ones((), device=dev), constraint=constraints.positive)
    pyro.sample("w", dist.Normal(wl, ws).to_event(1))
    pyro.sample("b", dist.Normal(bl, bs))
real = tmf._add_normal_kl
for graph in (True, False):
    for fused in (True, False):
        tmf._add_normal_kl = real if fused else (lambda *a: False)
        pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
        svi = SVI(examples.logreg_model, guide, pyro.optim.Adam({"lr": 0.01}), TraceMeanField_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1), hip_graph=graph, graph_warmup=2)
        for _ in range(10): svi.step(X, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 300 if graph else 100
        for _ in range(n): l = svi.step(X, y)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print("graph", graph, "fused_kl", fused, "us/step %.1f" % (dt * 1e6), "loss", float(l))
tmf._add_normal_kl = real
PY
