"""cProfile of the EAGER SVI.step of config 2 (developer tool): where the host time goes when the
user does not opt into hip_graph."""
import cProfile, pstats, sys, time
import torch
sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import SVI, Trace_ELBO
from pyro_amd.infer.autoguide import AutoNormal

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
X, y = examples.synthetic_logreg_data(N, 32, dev, seed=0)
pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
guide = AutoNormal(examples.logreg_model, init_scale=0.1)
svi = SVI(examples.logreg_model, guide, pyro.optim.Adam({"lr": 0.01}),
          Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1))
for _ in range(10):
    svi.step(X, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    svi.step(X, y)
torch.cuda.synchronize()
print("eager step: %.1f us" % ((time.perf_counter() - t0) / 200 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    svi.step(X, y)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
