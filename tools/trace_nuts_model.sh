#!/bin/bash
# kernel-by-kernel sequence of ONE tree round of NUTS on a MODEL potential (logistic regression through the
# handlers, jit_compile=True) + the whole-run leapfrog rate (developer tool): bash tools/trace_nuts_model.sh [N] [C]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
N=${1:-100000}; C=${2:-256}
OUT=gpurun_out/trace_nuts_model; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o t -- python tools/bench_nuts_model.py --n $N --chains $C --samples 60 --warmup 60 > $OUT/log.txt 2>&1
tail -5 $OUT/log.txt
python - "$OUT" <<'PY'
import csv, glob, sys, os, collections
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a round ends at a tree-advance kernel; show the last complete round and the gap statistics
idx = [i for i, n in enumerate(names) if "tree_advance" in n or "tree_run" in n]
lo, hi = idx[-3] + 1, idx[-2] + 1
t0 = int(rows[lo]["Start_Timestamp"])
with open(sys.argv[1] + "/one_round.txt", "w") as out:
    for r in rows[lo:hi]:
        out.write("%9.1f us  %7.1f us  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3,
                  (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:120]))
    spans = sorted((int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3 for a, b in zip(idx[-400:], idx[-399:]))
    out.write("round period over the last %d rounds: min %.1f, median %.1f, p90 %.1f us; %d launches per round\n"
              % (len(spans), spans[0], spans[len(spans) // 2], spans[int(len(spans) * .9)], hi - lo))
    tot = collections.Counter()
    for r in rows[idx[-401] + 1: idx[-1] + 1]:
        tot[r["Kernel_Name"][:80]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 / 400
    for k, v in tot.most_common(12):
        out.write("  %7.1f us/round  %s\n" % (v, k))
print(open(sys.argv[1] + "/one_round.txt").read())
os.remove(f)
PY
