export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/r02h
python tools/trace_models.py 2>&1 | grep "aten::\|====="
timeout 1500 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/r02h/gpu_tests.log 2>&1
tail -5 gpurun_out/r02h/gpu_tests.log
timeout 1200 python bench.py --no-others > gpurun_out/r02h/bench.json 2> gpurun_out/r02h/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02h/bench.json'))
for k in ['value','ms_per_step','steps']:
    print(k, json.dumps(d.get(k))[:900])
print({k: d['roofline'][k] for k in ('frac','kernel','kernel_ms','kernel_ms_rocprof','traffic','frac_bf16_mfma')})
print(d['secondary']['value'])
PY
