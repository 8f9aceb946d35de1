#!/bin/bash
# the round's measurement artefacts (copied into profiles/ afterwards): the bench line (driver arguments) + the full
# record, rocprofv3 stats + PMC + traffic of the headline, the graphed step's kernel sequence, per-kernel stats /
# one-step traces / traffic of configs 4 and 5, one round of NUTS on the model + the traffic of its GLM kernel, the
# tail's phase stamps.   usage: bash tools/collect_r06.sh [tag]
export TAG=${1:-r06}
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -s KILL 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_args.json 2> gpurun_out/${TAG}_bench.err
cp gpurun_out/bench_full.json gpurun_out/${TAG}_bench_full_driver_args.json 2>/dev/null
timeout -s KILL 600 python bench.py --full > gpurun_out/${TAG}_bench.json 2>> gpurun_out/${TAG}_bench.err
cp gpurun_out/bench_full.json gpurun_out/${TAG}_bench_full.json 2>/dev/null
timeout -s KILL 400 bash tools/prof.sh $TAG --no-nuts --no-model-nuts > gpurun_out/prof_$TAG.log 2>&1
GRAPHFLAG=" " timeout -s KILL 120 bash tools/trace_step.sh > gpurun_out/${TAG}_trace_step.txt 2>&1
timeout -s KILL 120 python tools/chain_stamps.py > gpurun_out/${TAG}_chain_stamps.txt 2>&1
for c in 4 5; do
  timeout -s KILL 150 bash tools/trace_cfg.sh $c > gpurun_out/${TAG}_cfg${c}_trace.txt 2>&1
  timeout -s KILL 150 bash tools/kstats_cfg.sh $c > gpurun_out/${TAG}_cfg${c}_kstats.txt 2>&1
  cp gpurun_out/kstats_cfg$c/kt/*kernel_stats.csv gpurun_out/${TAG}_cfg${c}_kernel_stats.csv 2>/dev/null
done
timeout -s KILL 300 bash tools/pmc_cfg.sh > gpurun_out/${TAG}_pmc_cfg.txt 2>&1
timeout -s KILL 120 bash tools/trace_nuts_model.sh 100000 256 > gpurun_out/${TAG}_nuts_model_round.txt 2>&1
timeout -s KILL 300 bash tools/nuts_model_traffic.sh > gpurun_out/${TAG}_nuts_model_traffic.log 2>&1
cut -c1-400 gpurun_out/${TAG}_bench_driver_args.json; echo; tail -2 gpurun_out/${TAG}_cfg4_trace.txt; tail -2 gpurun_out/${TAG}_cfg5_trace.txt; tail -4 gpurun_out/${TAG}_trace_step.txt
