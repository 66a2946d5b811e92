cd $GRAFT_REPO_ROOT
LIFE_FACTOR=1 LIFE_PROF=1 LIFE_SERIAL=1 python tools/life_check.py bls 1024 4096 20 1000 0 2>&1 | grep -v "^population\|reported\|objective factor" | head -16
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(1e3*d['timed_region_s'],2), round(d['roofline']['kernel_ms_per_launch'],2), d['roofline']['frac'])"; done
python -m pytest tests/test_gpu_life.py -m gpu -x -q -k "factored" 2>&1 | tail -2
