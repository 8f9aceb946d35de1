#!/bin/bash
# the round's measurement artefacts (copied into profiles/ afterwards): bench lines, rocprofv3 stats + PMC of the
# headline, the graphed step's kernel sequence, per-kernel stats / one-step traces / traffic of configs 4 and 5,
# one tree round of NUTS on the model + the traffic of its GLM kernel, the fuser's attribution
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -s KILL 400 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
timeout -s KILL 200 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_args.json 2>/dev/null
timeout -s KILL 400 bash tools/prof.sh r05 --no-nuts --no-model-nuts > gpurun_out/prof_r05.log 2>&1
GRAPHFLAG=" " timeout -s KILL 120 bash tools/trace_step.sh > gpurun_out/r05_trace_step.txt 2>&1
for c in 4 5; do
  timeout -s KILL 150 bash tools/trace_cfg.sh $c > gpurun_out/r05_cfg${c}_trace.txt 2>&1
  timeout -s KILL 150 bash tools/kstats_cfg.sh $c > gpurun_out/r05_cfg${c}_kstats.txt 2>&1
  cp gpurun_out/kstats_cfg$c/kt/*kernel_stats.csv gpurun_out/r05_cfg${c}_kernel_stats.csv 2>/dev/null
done
timeout -s KILL 300 bash tools/pmc_cfg.sh > gpurun_out/r05_pmc_cfg.txt 2>&1
timeout -s KILL 120 bash tools/trace_nuts_model.sh 100000 256 > gpurun_out/r05_nuts_model_round.txt 2>&1
timeout -s KILL 300 bash tools/nuts_model_traffic.sh > gpurun_out/r05_nuts_model_traffic.log 2>&1
timeout -s KILL 200 python tools/fuser_attribution.py > gpurun_out/r05_fuser_attribution.txt 2>&1
cut -c1-300 gpurun_out/r05_bench.json; echo; tail -2 gpurun_out/r05_cfg4_trace.txt; tail -2 gpurun_out/r05_cfg5_trace.txt; tail -4 gpurun_out/r05_trace_step.txt
