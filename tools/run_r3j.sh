#!/bin/bash
# scratch: bag-of-words forward at B = 1e5, then config 4
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -s KILL 300 python -m pytest tests/test_kernels_gpu.py tests/test_enum_gpu.py -x -q -m gpu -k 'bow or tsgemm or lda or tall or histogram or bag' 2>&1 | tail -3
timeout -s KILL 90 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -6
import sys; sys.path.insert(0, '.')
import torch
from pyro_amd import kernels as k
dev = torch.device('cuda:0')
for B in (3000, 100000):
    V, H, Wd = 1024, 100, 64
    g = torch.Generator(device='cpu').manual_seed(0)
    words = torch.randint(0, V, (Wd, B), generator=g).to(dev)
    ia, ib = k.bow_images(words, V)
    W = (torch.randn((H, V), generator=g) * 0.05).to(dev)
    bias = torch.randn((H,), generator=g).to(dev)
    counts = torch.zeros(V, B, device=dev).scatter_add(0, words, torch.ones(words.shape, device=dev))
    out = k.bow_linear_fwd(ia, W, bias, B)
    torch.cuda.synchronize()
    ref = counts.t() @ W.t() + bias
    print('fwd', B, 'max err', float((out - ref).abs().max()), flush=True)
    d0 = torch.randn((B, H), generator=g).to(dev)
    dW = k.bow_linear_bwd(ib, d0, V)
    print('bwd', B, 'max err', float((dW - d0.t() @ counts.t()).abs().max()), flush=True)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    d = torch.randn((B, H), generator=g).to(dev)
    for name, fn in (('fwd', lambda: k.bow_linear_fwd(ia, W, bias, B)), ('bwd', lambda: k.bow_linear_bwd(ib, d, V)),
                     ('tsgemm', lambda: k.tsgemm_tn(d, d))):
        fn(); s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        print('  ', name, B, '%.1f us per call (incl. split/reduce launches)' % (s.elapsed_time(e) * 100), flush=True)
PY
timeout -s KILL 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import sys; sys.path.insert(0, '.')
import torch
from tools import bench_configs as b
dev = torch.device('cuda:0')
r = b.config4(dev, steps=10)
print('config4:', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k != 'roofline'})
print(r.get('roofline'))
PY
