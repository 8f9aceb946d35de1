#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cat > /tmp/c4.py <<'PY'
import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import pyro_amd as pyro
from pyro_amd import examples
from pyro_amd.infer import SVI, TraceEnum_ELBO
dev = torch.device("cuda:0")
args = examples.LdaArgs(num_docs=100000)
data = examples.synthetic_lda_data(args, dev)
pyro.clear_param_store(); pyro.set_rng_seed(0); pyro.enable_validation(False)
predictor = examples.lda_make_predictor(args, dev)
guide = lambda data, args: examples.lda_guide(predictor, data, args)
svi = SVI(examples.lda_model, guide, pyro.optim.ClippedAdam({"lr": 0.01}), TraceEnum_ELBO(max_plate_nesting=2), hip_graph=False)
for _ in range(40): svi.step(data, args)
torch.cuda.synchronize()
PY
mkdir -p gpurun_out/c4prof
rm -rf gpurun_out/c4prof; mkdir -p gpurun_out/c4prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4prof -o c4 -- python /tmp/c4.py > /dev/null 2>&1
find gpurun_out/c4prof -name "*kernel_trace.csv" -delete
f=$(find gpurun_out/c4prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms per step", tot / 40 / 1e6)
for r in rows[:25]:
    print("%-90s calls/step %6.1f  us/step %8.1f  avg us %8.1f" % (r["Name"][:90], int(r["Calls"]) / 40, float(r["TotalDurationNs"]) / 40 / 1e3, float(r["AverageNs"]) / 1e3))
PY
