#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_enum_gpu.py -x -q -m gpu -p no:cacheprovider -k "bag_of_words or histogram or lda" > gpurun_out/r3f_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3f_tests.log; tail -12 gpurun_out/r3f_tests.log
timeout 600 python -c "
import torch, sys
sys.path.insert(0,'.')
from tools import bench_configs as bc
dev=torch.device('cuda:0')
print('config4', bc.config4(dev))
"
