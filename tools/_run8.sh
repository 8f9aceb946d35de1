export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/r02f
timeout 1200 python -m pytest tests -q -m gpu -x -p no:cacheprovider > gpurun_out/r02f/gpu_tests.log 2>&1
tail -8 gpurun_out/r02f/gpu_tests.log
timeout 1200 python bench.py > gpurun_out/r02f/bench.json 2> gpurun_out/r02f/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02f/bench.json'))
for k in ['value','ms_per_step','steps','roofline','cpu_baseline']:
    print(k, json.dumps(d.get(k))[:900])
print(json.dumps(d['secondary'].get('cpu_baseline'))[:600])
for k,v in d['other_configs'].items():
    print(k, {kk: vv for kk, vv in v.items() if kk != 'workload'} if isinstance(v, dict) else v)
PY
tail -3 gpurun_out/r02f/bench.err
