"""The captured config-2 step enqueued by ONE hipGraphLaunch against its two kernels launched one by one
(csrc/replay.hip), un-armed: every step() launches its own replay and waits for its loss (developer tool).

    python tools/graph_launch_host_cost.py [--clock]

Prints the host time of the enqueue call alone, and the step time of both forms in one process on the same
captured step, alternating.  CAUTION for whoever extends this: a replay that is enqueued WITHOUT reading its loss
leaves a publish in the mailbox that the next read_loss() takes for its own -- the host then runs ahead of the
device and every later "step" measures the device's cadence (GLM + tail back to back, ~63 us), not a step.  The
enqueue-only loops below re-synchronise the mailbox before anything else is timed."""
import sys
import time

import torch

sys.path.insert(0, ".")
import pyro_amd as pyro
from pyro_amd import examples, kernels
from pyro_amd.infer import SVI, Trace_ELBO
from pyro_amd.infer.autoguide import AutoNormal

kernels.DIRECT_REPLAY["on"] = True
dev = torch.device("cuda:0")
X, y = examples.synthetic_logreg_data(1_000_000, 32, dev, seed=0)
pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
if "--clock" in sys.argv:          # what bench.py adds: the GLM kernel's device-clock stamps
    clock = kernels.GlmDeviceClock(dev)
guide = AutoNormal(examples.logreg_model, init_scale=0.1)
svi = SVI(examples.logreg_model, guide, pyro.optim.Adam({"lr": 0.01}),
          Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1))
for _ in range(12):
    svi.step(X, y)
torch.cuda.synchronize()
entry = next(iter(svi._graphs.values()))
plan = entry.direct
print("captured: direct plan =", None if plan is None else plan.n_nodes)


def resync():
    torch.cuda.synchronize()
    entry._seq = int(entry._seq_np[0])


for name, call in (("hipGraphLaunch", entry.graph.replay), ("2 kernel launches", plan.launch if plan else None)):
    if call is None:
        continue
    for trial in range(2):
        resync()
        ts = []
        for i in range(8):
            t0 = time.perf_counter()
            call()
            ts.append((time.perf_counter() - t0) * 1e6)
        resync()
        print("%-18s enqueue, host us: %s" % (name, " ".join("%.1f" % t for t in ts)))

for rnd in range(3):
    for mode in ("direct", "graph"):
        if mode == "direct" and plan is None:
            continue
        entry.direct = plan if mode == "direct" else None
        resync()
        blocks = []
        for b in range(40):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(50):
                svi.step(X, y)
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / 50 * 1e6)
        blocks.sort()
        print("%-6s 40 x 50 un-armed steps: median block %.1f us/step (min %.1f, max %.1f)" % (
            mode, blocks[20], blocks[0], blocks[-1]), flush=True)
entry.direct = plan
