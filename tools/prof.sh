#!/bin/bash
# rocprofv3 recipe used for profiles/: kernel trace + stats, then separate PMC passes.
# usage (on the GPU box, from the repo root): bash tools/prof.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-others $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python bench.py $ARGS > $OUT/bench_kt.log 2>&1
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-others $* > $OUT/pmc_$N.log 2>&1
done
# keep only what is small: stats + per-kernel aggregates of the counter CSVs
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
summ = {}
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            agg[row["Kernel_Name"][:80]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            summ.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
    os.remove(f)
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
# HBM traffic of the dominant kernel, corrected as MI355X_MICROARCH.md (HBM section) prescribes:
# FETCH_SIZE (KB) counts half of a wide coalesced read on gfx950 -> x2; WRITE_SIZE as reported.
# The in-graph duration of the same kernel comes from the kernel-trace stats of the first pass.
kms, kcalls = None, -1
for f in glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        # the instantiation the captured step runs = the one with the most calls (eager warm-up and the
        # device-clock steps launch other instantiations a few times)
        if "glm_planes_" in row["Name"] and "kernel" in row["Name"] and "pack" not in row["Name"] \
                and int(row["Calls"]) > kcalls:
            kms, kcalls = float(row["AverageNs"]) / 1e6, int(row["Calls"])
for k, d in summ.items():
    if "glm_planes_" in k and "pack" not in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        f, w = d["FETCH_SIZE"]["mean"], d["WRITE_SIZE"]["mean"]
        json.dump({"kernel": k, "hbm_bytes_per_launch": (2 * f + w) * 1024, "kernel_ms_in_graph": kms,
                   "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w,
                   "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py; "
                          "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE reports "
                          "half of a wide coalesced read, MI355X_MICROARCH.md HBM section); "
                          "kernel_ms_in_graph = rocprofv3 --kernel-trace --stats average of the "
                          "graphed run; tools/prof.sh"}, open(out + "/traffic.json", "w"), indent=1)
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    os.remove(f)
for f in glob.glob(out + "/**/*.db", recursive=True):
    os.remove(f)
PY
du -sh $OUT
