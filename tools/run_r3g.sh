#!/bin/bash
# scratch: f16x2 plane image in the step
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_chain_gpu.py tests/test_svi_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3g_tests.txt
tail -5 gpurun_out/r3g_tests.txt
timeout 600 python bench.py > gpurun_out/r3g_bench.json 2> gpurun_out/r3g_bench.err
cat gpurun_out/r3g_bench.json | cut -c1-3000
tail -3 gpurun_out/r3g_bench.err
