#!/bin/bash
# scratch: bag-of-words kernels after the prefetch rewrite
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_enum_gpu.py -x -q -m gpu -k "bow or tsgemm or lda or tall" 2>&1 | tail -5
timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import sys; sys.path.insert(0, '.')
import torch
from tools import bench_configs as b
dev = torch.device('cuda:0')
r = b.config4(dev, steps=10)
print('config4:', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
PY
bash tools/trace_cfg.sh 4 > gpurun_out/r03_cfg4_trace.txt 2>&1; tail -2 gpurun_out/r03_cfg4_trace.txt
grep -E "bow_|tsgemm|Cijk|Fill" gpurun_out/r03_cfg4_trace.txt | cut -c1-120
