#!/bin/bash
# PMC passes over the two GLM kernels (developer tool): bash tools/pmc_glm.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
TAG=${1:-x}; OUT=gpurun_out/pmc_glm_$TAG; rm -rf $OUT; mkdir -p $OUT
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_TRANS"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -o b -- python tools/bench_glm_planes.py --pmc > $OUT/pmc_$N.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections, json
out = sys.argv[1]
summ = {}
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            summ.setdefault(k, {})[c] = sum(v) / len(v)
    os.remove(f)
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
for k, d in summ.items():
    if "glm" in k and "finalize" not in k and "pack" not in k:
        print(k)
        for c, v in sorted(d.items()):
            print("   %-30s %14.1f" % (c, v))
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True): os.remove(f)
for f in glob.glob(out + "/**/*.db", recursive=True): os.remove(f)
PY
