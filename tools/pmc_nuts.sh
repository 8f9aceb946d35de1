#!/bin/bash
# PMC counters of the fused NUTS kernel on the BASELINE configs[2] workload (developer tool)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
OUT=gpurun_out/pmc_nuts_${1:-x}; rm -rf $OUT; mkdir -p $OUT
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$N -o n -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-others > $OUT/$N.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections, os
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "nuts" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    os.remove(f)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    os.remove(f)
tot = {k: sum(v) for k, v in agg.items()}
print({k: round(v / 1024 / 12063184 * 1024, 1) for k, v in tot.items()}, "per leapfrog (assuming 12.06 M leapfrogs)")
PY
