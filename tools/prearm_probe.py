"""Why is (or is not) a default SVI step pre-armed? prints the gate's launch accounting (developer probe)."""
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
variants = {}
s0 = torch.ones((), device=gpu)
z32 = torch.zeros(32, device=gpu)
s32 = torch.ones(32, device=gpu)
z0 = torch.zeros((), device=gpu)


def m_scalar(X, y):
    w = pyro.sample("w", dist.Normal(z32, s0).to_event(1))
    b = pyro.sample("b", dist.Normal(z32[0], s0))
    with pyro.plate("data", X.shape[0]):
        pyro.sample("obs", dist.Bernoulli(logits=dist.linear_logits(X, w, b)), obs=y)


def m_exact(X, y):
    w = pyro.sample("w", dist.Normal(z32, s32).to_event(1))
    b = pyro.sample("b", dist.Normal(z0, s0))
    with pyro.plate("data", X.shape[0]):
        pyro.sample("obs", dist.Bernoulli(logits=dist.linear_logits(X, w, b)), obs=y)


pyro.enable_validation(False)
if "--eager-first" in sys.argv:
    pyro.clear_param_store()
    pyro.set_rng_seed(3)
    svi = SVI(m_scalar, AutoNormal(m_scalar, init_scale=0.1), pyro.optim.Adam({"lr": 0.02}),
              Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1), hip_graph=False)
    for _ in range(14):
        svi.step(X, y)
    print("eager run done", flush=True)
for name, model in (("bench", examples.logreg_model), ("scalar", m_scalar), ("exact", m_exact)):
    pyro.clear_param_store()
    pyro.set_rng_seed(3)
    svi = SVI(model, AutoNormal(model, init_scale=0.1), pyro.optim.Adam({"lr": 0.02}),
              Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1))
    for _ in range(8):
        svi.step(X, y)
    e = next(iter(svi._graphs.values()), None)
    g = getattr(e, "gate", None)
    print(name, "graphs", len(svi._graphs), "gate", g is not None, "armed", getattr(e, "armed", None),
          "reads", len(getattr(e, "reads", ())), "chain", svi.chain_stats, flush=True)
    svi.pause()
