#!/bin/bash
# round-3 GPU session C: in-kernel finalize: tests, stamps, A/B bench, trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_chain_gpu.py tests/test_svi_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r3c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r3c_tests.log; tail -8 gpurun_out/r3c_tests.log
timeout 300 python tools/chain_stamps.py 2>&1 | tail -3
for m in in_kernel separate; do
  echo "=== finalize $m"
  PYRO_AMD_GLM_FINALIZE=$m timeout 600 python bench.py --steps 300 --warmup 20 --no-nuts --no-others --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
done
GRAPHFLAG=" " timeout 600 bash tools/trace_step.sh 2>&1 | tail -5
