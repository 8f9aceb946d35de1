import sys, os
sys.path.insert(0, ".")
os.environ["PYRO_AMD_DEBUG_GRAPH"] = "1"
import torch, traceback
import pyro_amd as pyro
from pyro_amd import examples, kernels
from pyro_amd.infer import SVI, Trace_ELBO
from pyro_amd.infer.autoguide import AutoNormal
dev = torch.device("cuda:0")
N, D, G, P = 1_000_000, 32, 1000, 64
X, y, off = examples.synthetic_hier_logreg_data(N, D, G, dev, seed=0)
segs = kernels.GroupSegments(off, dev)
pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
guide = AutoNormal(examples.hier_logreg_model, init_scale=0.1)
svi = SVI(examples.hier_logreg_model, guide, pyro.optim.Adam({"lr": 0.01}),
          Trace_ELBO(num_particles=P, vectorize_particles=True, max_plate_nesting=1), hip_graph=True, graph_warmup=2)
try:
    for i in range(4):
        print(i, svi.step(X, y, segs))
except Exception:
    traceback.print_exc()
