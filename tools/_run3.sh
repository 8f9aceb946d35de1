export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/r02c
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "glm" -x -p no:cacheprovider 2>&1 | tail -5
timeout 600 python tools/bench_glm_planes.py 2>&1 | tee gpurun_out/r02c/bench_planes.log
bash tools/pmc_glm.sh r02c 2>&1 | grep -A30 "glm_planes_kernel"
