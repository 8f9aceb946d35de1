#!/bin/bash
# scratch: ring depth / occupancy sweep of the f16 plane-image kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "plane or planes or extreme" 2>&1 | tail -5
timeout 300 python tools/bench_glm_planes.py 2>&1 | grep -v amdgpu.ids
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, '.')
import torch
from tools import bench_configs as b
from pyro_amd import kernels as k
dev = torch.device('cuda:0')
for bpc in (2, 3, 4):
    k.glm_planes_tune(0, bpc)
    r = b.config5(dev, steps=30)
    print('config5 grouped wg/CU=%d: %.4f ms/step' % (bpc, r['ms_per_step']))
PY
