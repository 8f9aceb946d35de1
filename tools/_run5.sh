export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "glm_plane" -x -p no:cacheprovider 2>&1 | tail -3
for v in "" _noprio; do
  echo "== variant: base$v"
  ./tools/probes/glm_planes_probe$v | grep -E "^ring=|wave timeline|blocks" | head -22
done
timeout 600 python tools/bench_glm_planes.py 2>&1 | grep -v amdgpu.ids
