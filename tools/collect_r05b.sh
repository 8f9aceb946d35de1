#!/bin/bash
# second collection of round 5 (after the launch levels / map-reduce / fused activations): what changed since
# tools/collect_r05.sh -- the bench lines, the step traces and kernel stats of configs 4 / 5 and of examples/hmm.py,
# the traffic of configs 4 / 5, the fuser's attribution.  The headline kernel, its PMC summary and the NUTS rounds
# are those of collect_r05.sh (unchanged code).
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout -s KILL 500 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
timeout -s KILL 200 python bench.py --steps 20 --warmup 5 --no-others --no-cpu-baseline > gpurun_out/r05_bench_driver_args.json 2>/dev/null
for c in 4 5 h; do
  timeout -s KILL 200 bash tools/trace_cfg.sh $c > gpurun_out/r05_cfg${c}_trace.txt 2>&1
done
timeout -s KILL 150 bash tools/kstats_cfg.sh 4 > gpurun_out/r05_cfg4_kstats.txt 2>&1
cp gpurun_out/kstats_cfg4/kt/*kernel_stats.csv gpurun_out/r05_cfg4_kernel_stats.csv 2>/dev/null
timeout -s KILL 300 bash tools/pmc_cfg.sh > gpurun_out/r05_pmc_cfg.txt 2>&1
timeout -s KILL 300 python tools/fuser_attribution.py 4 5 1 h > gpurun_out/r05_fuser_attribution.txt 2>&1
cut -c1-300 gpurun_out/r05_bench.json; echo; for c in 4 5 h; do tail -1 gpurun_out/r05_cfg${c}_trace.txt; done
