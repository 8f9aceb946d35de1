#!/bin/bash
# kernel-by-kernel sequence of ONE graphed SVI.step of config 1 (eight schools; developer tool)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
OUT=gpurun_out/trace_cfg1; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o t -- python -c "
import sys; sys.path.insert(0,'.')
import torch
from tools import bench_configs as b
print(b.config1(torch.device("cuda:0"), steps=20))
" > $OUT/log.txt 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, os
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
lo, hi = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    print("%9.1f us  %7.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:120]))
os.remove(f)
PY
