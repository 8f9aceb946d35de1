export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/r02d
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "glm_plane" -x -p no:cacheprovider 2>&1 | tail -5
./tools/probes/glm_planes_probe
timeout 600 python tools/bench_glm_planes.py 2>&1 | tee gpurun_out/r02d/bench_planes.log
