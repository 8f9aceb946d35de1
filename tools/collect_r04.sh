#!/bin/bash
# the round's measurement artefacts (copied into profiles/ afterwards): bench line, rocprofv3 stats + PMC,
# the graphed step's kernel sequence, per-kernel stats and one-step traces of configs 4 and 5
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -s KILL 500 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
timeout -s KILL 120 python bench.py --steps 20 --warmup 5 --no-others --no-nuts --no-cpu-baseline > gpurun_out/r04_bench_driver_args.json 2>/dev/null
timeout -s KILL 200 python bench.py --steps 20000 --warmup 10 --no-others --no-nuts --no-cpu-baseline > gpurun_out/r04_bench_soak_20000_steps.json 2>/dev/null
timeout -s KILL 400 bash tools/prof.sh r04 --no-nuts > gpurun_out/prof_r04.log 2>&1
GRAPHFLAG=" " timeout -s KILL 120 bash tools/trace_step.sh > gpurun_out/r04_trace_step.txt 2>&1
for c in 4 5; do
  timeout -s KILL 150 bash tools/trace_cfg.sh $c > gpurun_out/r04_cfg${c}_trace.txt 2>&1
  timeout -s KILL 150 bash tools/kstats_cfg.sh $c > gpurun_out/r04_cfg${c}_kstats.txt 2>&1
  cp gpurun_out/kstats_cfg$c/kt/*kernel_stats.csv gpurun_out/r04_cfg${c}_kernel_stats.csv 2>/dev/null
done
cut -c1-400 gpurun_out/r04_bench.json; echo; tail -2 gpurun_out/r04_cfg4_trace.txt; tail -2 gpurun_out/r04_cfg5_trace.txt; tail -4 gpurun_out/r04_trace_step.txt
