#!/bin/bash
# HBM traffic of the NUTS kernels over one MCMC.run of bench.py's secondary workload (two PMC passes, as
# tools/prof.sh): writes gpurun_out/${TAG:-r06}_nuts_traffic.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
OUT=gpurun_out/nuts_pmc; rm -rf $OUT; mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o b -- python bench.py --steps 5 --warmup 5 --no-others --no-model-nuts --no-cpu-baseline > $OUT/$C.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(int)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(out + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "nuts" in k or "leapfrog" in k:
                tot[k[:70]][row["Counter_Name"]] += float(row["Counter_Value"])
                if row["Counter_Name"] == "FETCH_SIZE":
                    calls[k[:70]] += 1
secondary = None
for ln in open(out + "/FETCH_SIZE.log"):
    if ln.startswith("{"):
        secondary = json.loads(ln).get("secondary")
res = {"kernels": {k: {"launches": calls[k], "FETCH_SIZE_KB": v.get("FETCH_SIZE", 0.0), "WRITE_SIZE_KB": v.get("WRITE_SIZE", 0.0),
                       "hbm_bytes": (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024} for k, v in tot.items()},
       "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py's NUTS run, summed over every launch of the run; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, MI355X_MICROARCH.md)"}
res["hbm_bytes_total"] = sum(v["hbm_bytes"] for v in res["kernels"].values())
if secondary:
    res["leapfrogs"] = secondary.get("leapfrogs")
    if secondary.get("leapfrogs"):
        res["hbm_bytes_per_leapfrog"] = res["hbm_bytes_total"] / secondary["leapfrogs"]
json.dump(res, open("gpurun_out/%s_nuts_traffic.json" % os.environ.get("TAG", "r06"), "w"), indent=1)
print(json.dumps(res)[:1500])
PY
