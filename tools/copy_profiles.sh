#!/bin/bash
# gpurun_out/<tag>_* (written by tools/collect_r06.sh on the GPU box) -> profiles/, under the names bench.py's
# latest_profile() and DESIGN.md cite.   usage: bash tools/copy_profiles.sh [tag]
T=${1:-r06}; G=gpurun_out; P=profiles
cp $G/${T}_bench_driver_args.json $P/${T}_bench_driver_args.json
cp $G/${T}_bench_full_driver_args.json $P/${T}_bench_full_driver_args.json
cp $G/${T}_bench.json $P/${T}_bench.json
cp $G/${T}_bench_full.json $P/${T}_bench_full.json
cp $G/prof_$T/kt/*kernel_stats.csv $P/${T}_bench_kernel_stats.csv
cp $G/prof_$T/pmc_summary.json $P/${T}_pmc_summary.json
cp $G/prof_$T/traffic.json $P/${T}_traffic.json
cp $G/trace_step/last_step.txt $P/${T}_graphed_step_kernel_sequence.txt
grep "gap between" $G/${T}_trace_step.txt >> $P/${T}_graphed_step_kernel_sequence.txt
grep -v amdgpu.ids $G/${T}_chain_stamps.txt > $P/${T}_chain_tail_stamps.txt
for c in 4 5; do
  grep -v amdgpu.ids $G/${T}_cfg${c}_trace.txt > $P/${T}_cfg${c}_step_trace.txt
  cp $G/${T}_cfg${c}_kernel_stats.csv $P/${T}_cfg${c}_kernel_stats.csv
  cp $G/pmc_cfg/traffic_cfg$c.json $P/${T}_traffic_cfg$c.json
done
grep -v amdgpu.ids $G/${T}_nuts_model_round.txt > $P/${T}_nuts_model_round.txt
cp $G/${T}_nuts_model_traffic.json $P/${T}_nuts_model_traffic.json
ls -la $P/${T}_*
