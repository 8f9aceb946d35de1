#!/bin/bash
# HBM traffic of the dominant kernel of NUTS on the logistic-regression MODEL (the plane-image GLM kernel at
# P = chains), two PMC passes as tools/prof.sh: writes gpurun_out/${TAG:-r06}_nuts_model_traffic.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
OUT=gpurun_out/nuts_model_pmc; rm -rf $OUT; mkdir -p $OUT
for N in 100000 1000000; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/${N}_$C -o b -- python tools/bench_nuts_model.py --n $N --chains 256 --warmup 20 --samples 10 --depth 6 --no-compact > $OUT/${N}_$C.log 2>&1
  done
done
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections, os
out = sys.argv[1]
res = {}
for N in (100000, 1000000):
    agg = collections.defaultdict(list)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("%s/%d_%s/**/*counter_collection.csv" % (out, N, c), recursive=True):
            for row in csv.DictReader(open(f)):
                if "glm_planes_f16_kernel" in row["Kernel_Name"] and "pack" not in row["Kernel_Name"]:
                    agg[c].append(float(row["Counter_Value"]))
            os.remove(f)
    if agg["FETCH_SIZE"] and agg["WRITE_SIZE"]:
        # (the launches of full rounds: the gated ones at a span's end move nothing)
        f = sorted(agg["FETCH_SIZE"])[len(agg["FETCH_SIZE"]) // 2]
        w = sorted(agg["WRITE_SIZE"])[len(agg["WRITE_SIZE"]) // 2]
        res["N%d_C256" % N] = {"kernel": "glm_planes_f16_kernel, P = 256 chains", "FETCH_SIZE_KB_raw_median": f,
                               "WRITE_SIZE_KB_raw_median": w, "hbm_bytes_per_launch": (2 * f + w) * 1024,
                               "launches": len(agg["FETCH_SIZE"]), "algorithmic_bytes": N * (4 * 32 + 4)}
res["how"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/bench_nuts_model.py (full rounds, no "
              "compaction), median over the kernel's launches; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 "
              "correction of MI355X_MICROARCH.md); tools/nuts_model_traffic.sh")
json.dump(res, open("gpurun_out/%s_nuts_model_traffic.json" % os.environ.get("TAG", "r06"), "w"), indent=1)
print(json.dumps(res)[:1200])
for f in glob.glob(out + "/**/*.db", recursive=True): os.remove(f)
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True): os.remove(f)
PY
