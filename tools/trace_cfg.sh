#!/bin/bash
# kernel-by-kernel sequence of ONE graphed SVI.step of configs 4 / 5 (developer tool): bash tools/trace_cfg.sh 4|5|h
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
C=${1:-4}; OUT=gpurun_out/trace_cfg$C; rm -rf $OUT; mkdir -p $OUT
PA_NO_ROOFLINE=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o t -- python -c "
import sys; sys.path.insert(0,'.')
import torch
from tools import bench_configs as b
dev=torch.device('cuda:0')
print({'4': lambda: b.config4(dev, steps=4), '5': lambda: b.config5(dev, steps=4), 'h': lambda: b.config_hmm(dev, steps=3, graph=True)}['$C']())
" > $OUT/log.txt 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, os
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps repeat: the previous step ends where the last 8 kernel names of the trace occurred before
names = [r["Kernel_Name"] for r in rows]
m, end = 8, len(rows) - 1
prev = next(j for j in range(end - 1, m, -1) if names[j - m + 1:j + 1] == names[end - m + 1:end + 1])
lo, hi = prev + 1, end + 1
t0 = int(rows[lo]["Start_Timestamp"])
tot = {}
for r in rows[lo:hi]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%9.1f us  %7.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, d, r["Kernel_Name"][:110]))
    key = "pa::" if "pa::" in r["Kernel_Name"] else ("rocBLAS" if r["Kernel_Name"].startswith("Cijk") else "ATen/other")
    tot[key] = tot.get(key, 0.0) + d
span = (int(rows[hi - 1]["End_Timestamp"]) - t0) / 1e3
print("step span %.1f us, %d launches; kernel time by origin: %s" % (span, hi - lo, {k: round(v, 1) for k, v in tot.items()}))
os.remove(f)
PY
