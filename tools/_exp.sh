cd $GRAFT_REPO_ROOT
run() { python bench.py --gpus 1 --steps 20 --warmup 5 $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(1e3*d['timed_region_s'],2), round(d['roofline']['kernel_ms_per_launch'],2), d['value'], d['roofline']['frac'], d['roofline']['traffic_source'] if 'traffic_source' in d['roofline'] else '')"; }
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do run fixed "--no-cpu-baseline"; done
for i in 1 2; do run full ""; done
