#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
OUT=gpurun_out/trace_nm; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o t -- python tools/bench_nuts_model.py > $OUT/log.txt 2>&1
tail -3 $OUT/log.txt | grep jit
find $OUT -name "*kernel_trace.csv" -delete
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms", tot / 1e6, "kernels", sum(int(r["Calls"]) for r in rows))
for r in rows[:22]:
    print("%-86s calls %7s avg us %8.1f tot ms %8.1f" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
