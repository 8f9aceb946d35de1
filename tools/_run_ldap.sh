#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cat > /tmp/l.py <<'PY'
import sys, torch
sys.path.insert(0, "/root/repo")
from pyro_amd import kernels as k
dev = torch.device("cuda:0")
B, Wd, T, V = 100_000, 64, 8, 1024
g = torch.Generator(device="cpu").manual_seed(0)
words = torch.randint(0, V, (Wd, B), generator=g).to(dev)
lt = torch.log_softmax(torch.randn((B, T), generator=g), -1).to(dev)
lp = torch.log_softmax(torch.randn((T, V), generator=g), -1).to(dev)
index = k.lda_build_index(words, V)
for _ in range(20): k.lda_factor_fwd_bwd(words, lt, lp, index=index)
torch.cuda.synchronize()
PY
rm -rf gpurun_out/ldaprof; mkdir -p gpurun_out/ldaprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ldaprof -o l -- python /tmp/l.py > /dev/null 2>&1
find gpurun_out/ldaprof -name "*kernel_trace.csv" -delete
f=$(find gpurun_out/ldaprof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-100s calls %5s avg us %8.1f" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
