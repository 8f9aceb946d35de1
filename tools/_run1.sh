export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/r02a
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "glm_plane" -x -p no:cacheprovider > gpurun_out/r02a/planes_tests.log 2>&1
tail -15 gpurun_out/r02a/planes_tests.log
timeout 600 python tools/bench_glm_planes.py --more > gpurun_out/r02a/bench_planes.log 2>&1
cat gpurun_out/r02a/bench_planes.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02a/kt -o b -- python tools/bench_glm_planes.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r02a/kt/**/*kernel_stats.csv", recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:12]:
        print("%-90s %5s calls avg %8.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
import os
for g in glob.glob("gpurun_out/r02a/kt/**/*kernel_trace.csv", recursive=True): os.remove(g)
PY
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/r02a/gpu_tests.log 2>&1
tail -5 gpurun_out/r02a/gpu_tests.log
