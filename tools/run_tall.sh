#!/bin/bash
# developer tool: tests and timings of the tall-batch Linear kernels (csrc/tall.hip) and config 4
#   PA_TALL_WAVES=4|16 pins the waves per workgroup of tall_linear_kernel (default: by size)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -s KILL 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k 'tall or bow or histogram or bag' 2>&1 | tail -6
for NWV in 0; do
echo "PA_TALL_WAVES=$NWV"
PA_TALL_WAVES=$NWV timeout -s KILL 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -8
import sys; sys.path.insert(0, '.')
import torch
from pyro_amd import kernels as k
dev = torch.device('cuda:0')
B = 100000
g = torch.Generator(device='cpu').manual_seed(0)
x = torch.randn((B, 100), generator=g).to(dev); W = torch.randn((100, 100), generator=g).to(dev) * 0.1
b = torch.randn((100,), generator=g).to(dev); gr = torch.randn((B, 100), generator=g).to(dev)
W3 = torch.randn((8, 100), generator=g).to(dev); g3 = torch.randn((B, 8), generator=g).to(dev)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn in (('fwd 100->100', lambda: k.tall_linear(x, W, 1, 100, 100, b)),
                 ('dx  100->100', lambda: k.tall_linear(gr, W, 100, 1, 100)),
                 ('wgrad 100x100', lambda: k.tall_wgrad(gr, x)),
                 ('fwd 100->8', lambda: k.tall_linear(x, W3, 1, 100, 8, None)),
                 ('dx  8->100', lambda: k.tall_linear(g3, W3, 100, 1, 100)),
                 ('wgrad 8x100', lambda: k.tall_wgrad(g3, x))):
    fn(); s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    print('  %-14s %.1f us' % (name, s.elapsed_time(e) * 100), flush=True)
PY
PA_TALL_WAVES=$NWV timeout -s KILL 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import sys; sys.path.insert(0, '.')
import torch
from tools import bench_configs as b
dev = torch.device('cuda:0')
r = b.config4(dev, steps=10)
print('config4:', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k != 'roofline'})
PY
done
