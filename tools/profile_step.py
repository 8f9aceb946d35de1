"""Host-side profile of SVI.step on the config-2 workload (developer tool)."""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import SVI, Trace_ELBO
from pyro_amd.infer.autoguide import AutoNormal

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X, y = examples.synthetic_logreg_data(N, 32, dev)
pyro.enable_validation(False)
guide = AutoNormal(examples.logreg_model)
svi = SVI(examples.logreg_model, guide, pyro.optim.Adam({"lr": 0.01}),
          Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1))
for _ in range(5):
    svi.step(X, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    svi.step(X, y)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 20 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    svi.step(X, y)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
