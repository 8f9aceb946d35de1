"""Phase-boundary stamps of the LAST chained kernel of a captured config-5 step (developer tool; the step has
two chained kernels, the later one -- guide backward + Adam -- overwrites the stamps of the earlier)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyro_amd import kernels  # noqa: E402
from tools import bench_configs as bc  # noqa: E402

dev = torch.device("cuda:0")
stamps = torch.zeros(64, dtype=torch.int64, device=dev)
kernels.chain_debug_stamps(stamps)
r = bc.config5(dev, steps=10)
torch.cuda.synchronize()
print("ms/step", r["ms_per_step"])
s = stamps.cpu().tolist()
names = ["entry", "fin done", "multi: waited", "multi done", "mf: waited", "mf done", "adam: waited", "end"]
for base, who in ((0, "workgroup 0"), (16, "total wg")):
    t0 = s[base]
    print(who, " ".join("%s=%.2fus" % (n, (s[base + i] - t0) / 100.0) for i, n in enumerate(names) if s[base + i]))
inner = [v for v in s[32:62] if v]
if inner:
    print("in-body stamps of workgroup 0 (us from kernel entry):", " ".join("%.2f" % ((v - s[0]) / 100.0) for v in inner))
