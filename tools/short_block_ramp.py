"""Why are the first ~250 steps after a capture slower (developer tool)?  25 blocks of 20 captured config-2
steps between synchronisations, optionally after a busy HOST spin (argv[1] ms) and / or a DEVICE spin (argv[2]
ms of streaming load) -- prints the per-block microseconds per step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyro_amd as pyro  # noqa: E402
from pyro_amd import examples  # noqa: E402
from pyro_amd.infer import SVI, Trace_ELBO  # noqa: E402
from pyro_amd.infer.autoguide import AutoNormal  # noqa: E402

host_ms = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
dev_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
dev = torch.device("cuda:0")
X, y = examples.synthetic_logreg_data(1_000_000, 32, dev, seed=1)
pyro.set_rng_seed(0)
pyro.enable_validation(False)
svi = SVI(examples.logreg_model, AutoNormal(examples.logreg_model, init_scale=0.1), pyro.optim.Adam({"lr": lr}),
          Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1), hip_graph=True,
          graph_warmup=2)
for _ in range(5):
    svi.step(X, y)
torch.cuda.synchronize()
if dev_ms > 0:
    buf = torch.empty(64 << 20, dtype=torch.float32, device=dev)
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < dev_ms:
        for _ in range(8):
            buf.mul_(1.0)
        torch.cuda.synchronize()
if host_ms > 0:
    t = time.perf_counter()
    n = 0
    while (time.perf_counter() - t) * 1e3 < host_ms:
        n += 1
blocks = []
for _ in range(25):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        svi.step(X, y)
    torch.cuda.synchronize()
    blocks.append((time.perf_counter() - t0) / 20 * 1e6)
print("host spin %g ms, device spin %g ms, lr %g:" % (host_ms, dev_ms, lr), " ".join("%.1f" % b for b in blocks))
