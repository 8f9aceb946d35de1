#!/bin/bash
# per-kernel durations of the graphed bench step (developer tool): bash tools/kstats.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
TAG=${1:-k}; OUT=gpurun_out/kstats_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o b -- python bench.py --steps 40 --warmup 5 --no-nuts --no-others --no-cpu-baseline > $OUT/bench.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows[:28]:
    print("%-78s %5s calls  avg %8.1f us  %5.1f%%" % (r["Name"][:78], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
import os
for g in glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True): os.remove(g)
PY
tail -1 $OUT/bench.log | cut -c1-200
