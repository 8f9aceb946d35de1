#!/bin/bash
# the whole GPU suite, as the driver runs it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/gpu_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/gpu_tests.log; tail -15 gpurun_out/gpu_tests.log
timeout -s KILL 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
