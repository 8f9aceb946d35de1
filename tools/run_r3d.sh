#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_svi_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r3d_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3d_tests.log; tail -12 gpurun_out/r3d_tests.log
timeout 300 python tools/chain_stamps.py 2>&1 | tail -3
timeout 600 python bench.py --steps 300 --warmup 20 --no-nuts --no-others --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
GRAPHFLAG=" " timeout 600 bash tools/trace_step.sh 2>&1 | tail -6
