import sys, torch
sys.path.insert(0, ".")
from tools import bench_configs as bc
from pyro_amd import kernels as k
dev = torch.device("cuda:0")
for bpc in (4, 3, 2):
    k.glm_planes_tune(0, bpc)
    r = bc.config5(dev, steps=20)
    print("bpc", bpc, round(r["ms_per_step"], 4), "kernel_ms", r["roofline"]["kernel_ms"])
