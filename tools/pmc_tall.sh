#!/bin/bash
# developer tool: PMC of the tall-batch Linear kernels (forward 100 -> 100 over 1e5 rows), two passes
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
OUT=gpurun_out/pmc_tall; rm -rf $OUT; mkdir -p $OUT
cat > $OUT/run.py <<'PY'
import sys; sys.path.insert(0, '.')
import torch
from pyro_amd import kernels as k
dev = torch.device('cuda:0')
B = 100000
g = torch.Generator(device='cpu').manual_seed(0)
x = torch.randn((B, 100), generator=g).to(dev); W = torch.randn((100, 100), generator=g).to(dev) * 0.1
b = torch.randn((100,), generator=g).to(dev); gr = torch.randn((B, 100), generator=g).to(dev)
for _ in range(6):
    k.tall_linear(x, W, 1, 100, 100, b); k.tall_wgrad(gr, x)
torch.cuda.synchronize()
PY
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | cut -c1-20 | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$N -o t -- python $OUT/run.py > $OUT/$N.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "tall_" in row["Kernel_Name"]:
            agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s %14.0f" % (c, sum(v) / len(v)))
PY
