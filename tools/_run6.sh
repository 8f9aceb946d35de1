export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/r02e
timeout 900 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/r02e/gpu_tests.log 2>&1
tail -8 gpurun_out/r02e/gpu_tests.log
timeout 900 python bench.py --steps 50 --warmup 5 > gpurun_out/r02e/bench.json 2> gpurun_out/r02e/bench.err
tail -c 3000 gpurun_out/r02e/bench.json; tail -5 gpurun_out/r02e/bench.err
