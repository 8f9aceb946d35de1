export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_svi_gpu.py tests/test_mcmc_gpu.py -q -m gpu -x -p no:cacheprovider -k "glm or north_star or reference_model or deferred or jit_compile" --durations=5 2>&1 | tail -15
