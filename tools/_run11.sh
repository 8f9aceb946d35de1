export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_enum_gpu.py -q -m gpu -x -p no:cacheprovider -k "marginals or posterior" 2>&1 | tail -6
