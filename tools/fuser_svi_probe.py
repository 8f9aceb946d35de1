"""Which part of a captured SVI step does not get along with generated kernels? (developer probe)"""
import sys
import torch
sys.path.insert(0, ".")
from tools import bench_configs as bc
from pyro_amd.ops import fuser
dev = torch.device("cuda:0")
r = bc.config1(dev, steps=20)
print({k: r[k] for k in ("us_per_step", "graphed", "last_loss")}, fuser.STATS, flush=True)
