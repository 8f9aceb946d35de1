cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout -s KILL 400 python -m pytest tests/test_multisite_gpu.py tests/test_kernels_gpu.py tests/test_svi_gpu.py -x -q -m gpu -p no:cacheprovider -k "closed_form or sum_to_nd or meanfield_score or hier" 2>&1 | tail -12
timeout -s KILL 200 bash tools/trace_cfg.sh 5 > gpurun_out/cfg5_trace_now.txt 2>&1; cut -c1-130 gpurun_out/cfg5_trace_now.txt | tail -32
timeout -s KILL 200 python - <<'PY' 2>&1 | tail -3
import sys, torch
sys.path.insert(0, ".")
from tools import bench_configs as bc
dev = torch.device("cuda:0")
r = bc.config5(dev, steps=30)
print("cfg5 ms/step", round(r["ms_per_step"], 4), "kernel_ms", r["roofline"]["kernel_ms"])
PY
