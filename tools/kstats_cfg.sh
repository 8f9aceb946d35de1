#!/bin/bash
# per-kernel durations of configs 4 / 5 (developer tool): bash tools/kstats_cfg.sh 5|4|mvn|p1|hmm
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
C=${1:-5}; OUT=gpurun_out/kstats_cfg$C; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o b -- python -c "
import sys; sys.path.insert(0,'.')
import torch
from tools import bench_configs as b
dev=torch.device('cuda:0')
print({'5': lambda: b.config5(dev, steps=10), '4': lambda: b.config4(dev, steps=5),
       'mvn': lambda: b.config2_variant(dev, 'mvn', steps=20), 'p1': lambda: b.config2_variant(dev, 'normal', P=1, steps=20),
       'hmm': lambda: b.config_hmm(dev, steps=5, graph=True),
       'hmmv': lambda: b.config_hmm_vectorised(dev, steps=10)}['$C']())
" > $OUT/log.txt 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, os
f = glob.glob(sys.argv[1] + "/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print("%-84s %5s calls  avg %9.1f us  %5.1f%%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
for g in glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True): os.remove(g)
PY
tail -1 $OUT/log.txt | cut -c1-200
