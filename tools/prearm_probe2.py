"""Final parameters of a pre-armed run against an eager run (developer probe)."""
import sys
import torch
sys.path.insert(0, ".")
import pyro_amd as pyro
import pyro_amd.distributions as dist
from pyro_amd import examples
from pyro_amd.infer import SVI, Trace_ELBO
from pyro_amd.infer.autoguide import AutoNormal

gpu = torch.device("cuda:0")
X, y = examples.synthetic_logreg_data(20000, 32, gpu, seed=2)
prior_scale = torch.ones((), device=gpu)
zeros = torch.zeros(32, device=gpu)


def model(X, y):
    w = pyro.sample("w", dist.Normal(zeros, prior_scale).to_event(1))
    b = pyro.sample("b", dist.Normal(zeros[0], prior_scale))
    with pyro.plate("data", X.shape[0]):
        pyro.sample("obs", dist.Bernoulli(logits=dist.linear_logits(X, w, b)), obs=y)


pyro.enable_validation(False)
nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 14
res = {}
for mode in ("eager", "graph", "auto", "auto+pause"):
    pyro.clear_param_store()
    pyro.set_rng_seed(3)
    kw = {"eager": dict(hip_graph=False), "graph": dict(hip_graph=True), "auto": {}, "auto+pause": {}}[mode]
    svi = SVI(model, AutoNormal(model, init_scale=0.1), pyro.optim.Adam({"lr": 0.02}),
              Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1), **kw)
    losses = [svi.step(X, y) for _ in range(nsteps)]
    if mode == "auto+pause":
        svi.pause()
        torch.cuda.synchronize()
    res[mode] = (losses, {k: v.detach().clone() for k, v in pyro.get_param_store().items()})
    svi.pause()
    torch.cuda.synchronize()
for mode in ("graph", "auto", "auto+pause"):
    print(mode, "losses equal:", res[mode][0] == res["eager"][0],
          {k: float((res[mode][1][k] - res["eager"][1][k]).abs().max()) for k in res["eager"][1]}, flush=True)
