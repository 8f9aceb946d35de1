"""RCCL at world size 1 on a one-GPU box (developer tool): the flat-gradient all-reduce as an eager
launch between two graphs (the default multi-rank step) and captured INSIDE the step's graph
(PYRO_AMD_GRAPH_COLLECTIVE=1), against the plain single-process step: same losses, and what each
costs.  Run under `timeout`."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import SVI, Trace_ELBO
from pyro_amd.infer.autoguide import AutoNormal

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
N_ROWS = int(os.environ.get("RCCL_ONE_RANK_ROWS", "1000000"))
N_STEPS = int(os.environ.get("RCCL_ONE_RANK_STEPS", "200"))
X, y = examples.synthetic_logreg_data(N_ROWS, 32, dev, seed=0)


def run(mode):
    pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
    optim = pyro.optim.Adam({"lr": 0.01})
    if mode != "plain":
        optim = pyro.optim.RcclOptimizer(optim)
        optim.force_collective = True
    os.environ["PYRO_AMD_GRAPH_COLLECTIVE"] = "1" if mode == "one_graph" else "0"
    g = AutoNormal(examples.logreg_model, init_scale=0.1)
    svi = SVI(examples.logreg_model, g, optim, Trace_ELBO(num_particles=64, vectorize_particles=True,
                                                         max_plate_nesting=1),
              hip_graph=(mode != "eager"), graph_warmup=2)
    losses = [svi.step(X, y) for _ in range(8)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = N_STEPS
    for _ in range(n):
        svi.step(X, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    ent = next(iter(svi._graphs.values()), None) if svi.hip_graph else None
    shape = "eager" if ent is None else ("two graphs + eager collective" if ent.graph2 is not None else "one graph")
    print("%-10s us/step %7.1f  step structure: %-30s losses %s" % (mode, dt * 1e6, shape, [round(x, 1) for x in losses[:4]]), flush=True)
    return losses


ref = run("plain")
for mode in ("eager", "split", "one_graph"):
    got = run(mode)
    assert all(abs(a - b) <= 1e-6 * abs(b) for a, b in zip(got, ref)), (mode, got, ref)
print("RCCL one-rank OK")
dist.destroy_process_group()
