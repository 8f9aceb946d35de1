"""What a captured step of configs 4 / 5 / 1 hands to the fuser and what it still runs as separate ATen
operators (developer tool): per config the step time with the fuser off / on, operators recorded, kernels
generated, and the operators met but not taken.    python tools/fuser_attribution.py [4 5 1]"""
import os
import sys

import torch

sys.path.insert(0, ".")
os.environ["PA_NO_ROOFLINE"] = "1"
from pyro_amd.ops import fuser  # noqa: E402
from tools import bench_configs as bc  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1:] or ["4", "5", "1"]
run = {"4": lambda: bc.config4(dev, steps=10), "5": lambda: bc.config5(dev, steps=20),
       "1": lambda: bc.config1(dev, steps=100)}
for c in which:
    for on in (False, True):
        fuser.ENABLED["on"] = on
        fuser.UNFUSED.clear()
        b = dict(fuser.STATS)
        r = run[c]()
        t = r.get("ms_per_step", r.get("us_per_step", 0) / 1e3)
        d = {k: fuser.STATS[k] - b[k] for k in fuser.STATS}
        print("config %s fuser %-5s %.4f ms/step graphed=%s %s" % (c, on, t, r["graphed"], d), flush=True)
        if on:
            print("   not taken:", dict(sorted(fuser.UNFUSED.items(), key=lambda kv: -kv[1])), flush=True)
