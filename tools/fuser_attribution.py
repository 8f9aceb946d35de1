"""What a captured step of configs 4 / 5 / 1 hands to the fuser and what it still runs as separate ATen
operators (developer tool): per config the step time with the fuser off / on, operators recorded, kernels
generated, and the operators met but not taken.    python tools/fuser_attribution.py [4 5 1] [--why]
--why: also where every operator that was not taken comes from (innermost frames) and what each launch holds."""
import os
import sys

import torch

sys.path.insert(0, ".")
os.environ["PA_NO_ROOFLINE"] = "1"
from pyro_amd.ops import fuser  # noqa: E402
from tools import bench_configs as bc  # noqa: E402

dev = torch.device("cuda:0")
why = "--why" in sys.argv
which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["4", "5", "1"]
run = {"4": lambda: bc.config4(dev, steps=10), "5": lambda: bc.config5(dev, steps=20),
       "1": lambda: bc.config1(dev, steps=100), "h": lambda: bc.config_hmm(dev, steps=5, graph=True)}
for c in which:
    for on in (False, True):
        fuser.ENABLED["on"] = on
        fuser.UNFUSED.clear()
        fuser.TRACE.update(on=why and on, sites={}, kernels=[])
        b = dict(fuser.STATS)
        r = run[c]()
        t = r.get("ms_per_step", r.get("us_per_step", 0) / 1e3)
        d = {k: fuser.STATS[k] - b[k] for k in fuser.STATS}
        print("config %s fuser %-5s %.4f ms/step graphed=%s loss=%r %s" % (
            c, on, t, r["graphed"], r.get("loss_last", r.get("last_loss")), d), flush=True)
        if on:
            print("   not taken:", dict(sorted(fuser.UNFUSED.items(), key=lambda kv: -kv[1])), flush=True)
            if why:
                for name, sites in fuser.TRACE["sites"].items():
                    for where, shapes in sites[:len(sites) // 2 or 1]:
                        print("      %-28s %s   %s" % (name, shapes, where))
                body = 0
                for rec in fuser.TRACE["kernels"][:len(fuser.TRACE["kernels"]) // 2]:
                    if rec[0] == "launch":
                        print("      ---- launch of %d bodies" % rec[1])
                    else:
                        print("      body %s: %s" % (rec[0], " ".join("%s%s" % (o, "" if lv else "'") for o, _, lv in rec[1])))
