#!/bin/bash
# the whole GPU suite with the HIP runtime's error log on (developer tool)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
AMD_LOG_LEVEL=1 timeout -s KILL 900 python -X faulthandler -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/gpu_tests_dbg.log 2>&1
echo "tests rc=$?" >> gpurun_out/gpu_tests_dbg.log
grep -v "^\.\|^$" gpurun_out/gpu_tests_dbg.log | grep -v "site-packages\|pluggy\|_pytest" | tail -40
