export PYTHONDONTWRITEBYTECODE=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt_lda
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_lda -o b -- python tools/bench_lda.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
f = glob.glob("gpurun_out/kt_lda/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:10]:
        print("%-100s %5s calls avg %8.1f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
for g in glob.glob("gpurun_out/kt_lda/**/*kernel_trace.csv", recursive=True): os.remove(g)
PY
