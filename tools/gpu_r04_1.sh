cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -s KILL 900 python -m pytest tests/test_group_rows_gpu.py tests/test_svi_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/g1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/g1_tests.log; tail -25 gpurun_out/g1_tests.log
timeout -s KILL 300 python - > gpurun_out/g1_cfg5.log 2>&1 <<'PY'
import torch, sys
sys.path.insert(0, ".")
from tools import bench_configs as bc
dev = torch.device("cuda:0")
print("cfg5 reference text:", bc.config5(dev))
print("cfg5 sorted/grouped spelling:", bc.config5(dev, reference_text=False))
PY
tail -5 gpurun_out/g1_cfg5.log
