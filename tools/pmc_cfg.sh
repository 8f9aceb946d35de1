#!/bin/bash
# HBM traffic of the dominant kernels of configs 4 and 5 (separate --pmc passes, as tools/prof.sh):
# bash tools/pmc_cfg.sh  ->  gpurun_out/pmc_cfg/traffic_cfg{4,5}.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
OUT=gpurun_out/pmc_cfg; rm -rf $OUT; mkdir -p $OUT
for C in 4 5; do
  for CTR in FETCH_SIZE WRITE_SIZE; do
    PA_NO_ROOFLINE=1 timeout -s KILL 200 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT/c${C}_$CTR -o b -- python -c "
import sys; sys.path.insert(0,'.')
import torch
from tools import bench_configs as b
dev=torch.device('cuda:0')
print({'4': lambda: b.config4(dev, steps=3), '5': lambda: b.config5(dev, steps=3)}['$C']())
" > $OUT/c${C}_$CTR.log 2>&1
  done
done
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections, os
out = sys.argv[1]
want = {"4": ("bow_linear_fwd_kernel", "bow_linear_bwd_kernel", "lda_vocab_kernel", "tall_linear_kernel", "tall_wgrad_kernel"),
        "5": ("glm_planes_f16_kernel", "glm_planes_kernel", "meanfield_sample_kernel")}
for c, names in want.items():
    res = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("%s/c%s_%s/**/*counter_collection.csv" % (out, c, ctr), recursive=True):
            agg = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                for n in names:
                    if n in row["Kernel_Name"] and "pack" not in row["Kernel_Name"]:
                        agg[n].append(float(row["Counter_Value"]))
            for n, v in agg.items():
                res.setdefault(n, {})[ctr + "_KB_raw"] = sum(v) / len(v)
            os.remove(f)
    for n, d in res.items():
        if "FETCH_SIZE_KB_raw" in d and "WRITE_SIZE_KB_raw" in d:
            d["hbm_bytes_per_launch"] = (2 * d["FETCH_SIZE_KB_raw"] + d["WRITE_SIZE_KB_raw"]) * 1024
    res["how"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/bench_configs.config%s; HBM bytes = "
                  "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction of MI355X_MICROARCH.md); tools/pmc_cfg.sh" % c)
    json.dump(res, open("%s/traffic_cfg%s.json" % (out, c), "w"), indent=1)
    print(c, {n: round(d.get("hbm_bytes_per_launch", 0) / 1e6, 1) for n, d in res.items() if isinstance(d, dict)})
for f in glob.glob(out + "/**/*.db", recursive=True): os.remove(f)
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True): os.remove(f)
PY
