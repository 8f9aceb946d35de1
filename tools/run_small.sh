#!/bin/bash
# developer tool: tests of the small-operator interpreter, then configs 4 / 5 with it on and off
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -s KILL 300 python -m pytest tests/test_smallops_gpu.py -x -q -m gpu 2>&1 | tail -25
for flag in 1 0; do
PYRO_AMD_SMALLOPS=$flag PA_NO_ROOFLINE=1 timeout -s KILL 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import os, sys; sys.path.insert(0, '.')
import torch
from tools import bench_configs as b
dev = torch.device('cuda:0')
r4 = b.config4(dev, steps=10); r5 = b.config5(dev, steps=20)
print('SMALLOPS=%s  config4 %.4f ms %s  config5 %.4f ms %s' % (os.environ['PYRO_AMD_SMALLOPS'], r4['ms_per_step'], r4['smallops_launches_recorded'], r5['ms_per_step'], r5['smallops_launches_recorded']))
PY
done
