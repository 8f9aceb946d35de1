#!/bin/bash
# PMC profile of the GLM kernels alone (developer tool): bash tools/prof_glm.sh <tag>
set -u
TAG=${1:-glm}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python tools/bench_glm.py --quick"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o glm -- $CMD > $OUT/kt.log 2>&1
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -o glm -- $CMD > $OUT/pmc_$N.log 2>&1
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
            summ.setdefault(k, {})[c] = {"mean": sum(v) / len(v), "n": len(v)}
    os.remove(f)
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1, sort_keys=True)
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    os.remove(f)
for f in glob.glob(out + "/**/*.db", recursive=True):
    os.remove(f)
for k, d in summ.items():
    if "glm_bernoulli" in k:
        print(k)
        for c in sorted(d):
            print("   %-28s %.4g" % (c, d[c]["mean"]))
PY
head -5 $OUT/kt/glm_kernel_stats.csv | cut -c1-200
