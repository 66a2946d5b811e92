mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_life.py -q -m gpu -x -k "dispatch or round4 or launcher" 2>&1 | tail -4
timeout 900 python bench.py --steps 20 --warmup 5 --no-secondary > gpurun_out/bench_r05_b.json 2> gpurun_out/bench_r05_b.err; tail -c 400 gpurun_out/bench_r05_b.err
python - <<'PY'
import json
txt=open('gpurun_out/bench_r05_b.json').read()
d=json.loads([l for l in txt.splitlines() if l.startswith('{')][-1])
print({k:d[k] for k in ('value','ms_per_step','n_gpus','steps')}, d['roofline']['kernel'], round(d['roofline']['frac'],4), d['roofline']['traffic_provenance'])
print('other:', d['schemes'].get('other_lifecycle_kernel'))
print('two:', d['schemes'].get('two',{}).get('value'))
print('cpu', d.get('cpu_baseline',{}).get('value'), d['best'])
PY
