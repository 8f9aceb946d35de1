#!/bin/bash
# round-3 GPU session A: chain + hoisting tests, graphed-step trace, short bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_svi_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r3a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3a_tests.log
tail -15 gpurun_out/r3a_tests.log
GRAPHFLAG=" " timeout 600 bash tools/trace_step.sh > gpurun_out/r3a_trace.log 2>&1
tail -12 gpurun_out/r3a_trace.log
timeout 600 python bench.py --steps 300 --warmup 20 --no-nuts --no-others --no-cpu-baseline > gpurun_out/r3a_bench.log 2>&1
tail -1 gpurun_out/r3a_bench.log | cut -c1-400
timeout 300 python tools/chain_stamps.py > gpurun_out/r3a_stamps.log 2>&1; tail -4 gpurun_out/r3a_stamps.log
