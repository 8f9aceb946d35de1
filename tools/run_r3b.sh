#!/bin/bash
# round-3 GPU session B: NT-hint A/B of the image loads; chain stamps; bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for v in "" "_nont"; do
  export PYRO_AMD_LIB=$PWD/pyro_amd/lib/libpyro_amd$v.so
  echo "=== lib$v"
  timeout 300 python tools/chain_stamps.py 2>&1 | tail -3
  timeout 600 python bench.py --steps 300 --warmup 20 --no-nuts --no-others --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
done
unset PYRO_AMD_LIB
GRAPHFLAG=" " timeout 600 bash tools/trace_step.sh 2>&1 | tail -5
