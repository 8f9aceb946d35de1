"""Phase-boundary stamps of the chained tail in the captured config-2 step (developer tool):
python tools/chain_stamps.py [N] -> microseconds from kernel entry, workgroup 0 and the total's."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyro_amd as pyro  # noqa: E402
from pyro_amd import examples, kernels  # noqa: E402
from pyro_amd.infer import SVI, Trace_ELBO  # noqa: E402
from pyro_amd.infer.autoguide import AutoNormal  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
X, y = examples.synthetic_logreg_data(N, 32, dev, seed=1)
pyro.set_rng_seed(0)
pyro.enable_validation(False)
stamps = torch.zeros(64, dtype=torch.int64, device=dev)
kernels.chain_debug_stamps(stamps)
svi = SVI(examples.logreg_model, AutoNormal(examples.logreg_model, init_scale=0.1),
          pyro.optim.Adam({"lr": 0.01}),
          Trace_ELBO(num_particles=64, vectorize_particles=True, max_plate_nesting=1), hip_graph=True)
for i in range(30):
    if i == 29:
        stamps.zero_()
        kernels.chain_debug_stamps(stamps)      # (re-arms the in-body stamp counter)
    loss = svi.step(X, y)
torch.cuda.synchronize()
print("chain stats", svi.chain_stats, "loss", loss)
s = stamps.cpu().tolist()
names = ["entry", "fin done", "waited for fin", "extras done", "site backward done", "site adam done", "arrived", "pre-wait work done"] if svi.chain_fused else ["entry", "fin done", "multi: waited", "multi done", "mf: waited", "mf done", "adam: waited", "end"]
for base, who in ((0, "workgroup 0"), (16, "total wg")):
    t0 = s[base]
    print(who, " ".join("%s=%.2fus" % (n, (s[base + i] - t0) / 100.0) for i, n in enumerate(names) if s[base + i]))
if s[24]:
    print("latency probe (site workgroup 0, thread 0): kernarg table read %.2fus, dependent entry read +%.2fus, data read +%.2fus, second data read +%.2fus"
          % ((s[24] - s[0]) / 100.0, (s[25] - s[24]) / 100.0, (s[26] - s[25]) / 100.0, (s[27] - s[26]) / 100.0))
inner = [v for v in s[32:62] if v]
if inner:
    print("in-body stamps of workgroup 0 (us from kernel entry):", " ".join("%.2f" % ((v - s[0]) / 100.0) for v in inner))
