mkdir -p gpurun_out
for w in 5 20 60; do python bench.py --steps 20 --warmup $w --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('warmup', r['warmup'], 'ms/step', r['ms_per_step'], 'frac', r['roofline']['frac'], 'kernel ms', r['roofline']['kernel_ms_per_launch'])"; done
