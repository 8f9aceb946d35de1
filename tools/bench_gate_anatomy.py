"""Where a pre-armed step's time goes (developer tool): host stamps at the release (store to the gate's
go word) and at the moment the loss is seen."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import SVI, Trace_ELBO, svi as svi_mod
from pyro_amd.infer.autoguide import AutoNormal

dev = torch.device("cuda:0")
X, y = examples.synthetic_logreg_data(1_000_000, 32, dev, seed=0)
pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
guide = AutoNormal(examples.logreg_model, init_scale=0.1)
svi = SVI(examples.logreg_model, guide, pyro.optim.Adam({"lr": 0.01}),
          Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1), hip_graph=True,
          graph_warmup=2, prearm=True)
for _ in range(8):
    svi.step(X, y)
(entry,) = svi._graphs.values()
assert entry.gate is not None and entry.armed
stamps = []
orig_launch, orig_read = entry.launch, entry.read_loss
def launch():
    r = orig_launch(); stamps.append(("go", time.perf_counter())); return r
def read_loss(released_armed=False):
    v = orig_read(released_armed); stamps.append(("loss", time.perf_counter())); return v
entry.launch, entry.read_loss = launch, read_loss
torch.cuda.synchronize()
for _ in range(400):
    svi.step(X, y)
t = [s for s in stamps[100:]]
go = np.array([v for k, v in t if k == "go"]); loss = np.array([v for k, v in t if k == "loss"])
n = min(len(go), len(loss))
print("release -> loss seen   median %.1f us" % (np.median(loss[:n] - go[:n]) * 1e6))
print("loss seen -> next release median %.1f us" % (np.median(go[1:n] - loss[:n - 1]) * 1e6))
print("step cadence           median %.1f us" % (np.median(np.diff(go[:n])) * 1e6))
