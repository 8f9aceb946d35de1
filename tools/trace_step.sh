#!/bin/bash
# kernel-by-kernel sequence of ONE eager SVI.step of the bench workload (developer tool)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
OUT=gpurun_out/trace_step; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o t -- python bench.py --steps 40 --warmup 8 ${GRAPHFLAG:---no-graph} --no-nuts --no-model-nuts --no-others --no-cpu-baseline --plate ${1:-1000000} > $OUT/log.txt 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = after the last but one adam kernel
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"] or "chain_" in r["Kernel_Name"]]
# the step of MEDIAN span among the replays (the tightest one is a replay its gate gave up -- every node of
# it returns at once -- and the widest an eager warm-up step)
spans = sorted((int(rows[b]["End_Timestamp"]) - int(rows[a + 1]["Start_Timestamp"]), a + 1, b + 1)
               for a, b in zip(idx, idx[1:]))
best = spans[len(spans) // 2]
lo, hi = best[1], best[2]
t0 = int(rows[lo]["Start_Timestamp"])
with open(sys.argv[1] + "/last_step.txt", "w") as out:
    for r in rows[lo:hi]:
        line = "%9.1f us  %7.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:110])
        out.write(line + "\n")
print(open(sys.argv[1] + "/last_step.txt").read())
# idle time of the GPU between consecutive steps (end of a step's last kernel -> start of the next
# step's first kernel), over the replays
gaps = [(int(rows[b + 1]["Start_Timestamp"]) - int(rows[b]["End_Timestamp"])) / 1e3 for b in idx[:-1] if b + 1 < len(rows)]
if gaps:
    gaps.sort()
    print("gap between steps: min %.1f us, median %.1f us over %d steps" % (gaps[0], gaps[len(gaps) // 2], len(gaps)))
import os; os.remove(f)
PY
