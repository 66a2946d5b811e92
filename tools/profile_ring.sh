#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel statistics of the headline bench in the ring scheme (ONE persistent
# slot-queue launch).  Kernel trace only: counter collection serialises dispatches, which a persistent launch that waits for
# the kernels of other streams cannot survive.  Output: gpurun_out/prof_ring_$TAG/.
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_ring_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python $ROOT/bench.py --scheme ring --no-secondary --no-cpu-baseline > $OUT/stats.log 2>&1
grep "^{" $OUT/stats.log > $OUT/bench_line.json
find $OUT -name "*kernel_trace.csv" -delete < /dev/null
find $OUT -name "*kernel_stats.csv" -exec head -8 {} \; < /dev/null | cut -c1-170
python -c "
import json; d=json.load(open('$OUT/bench_line.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['persistent_launch'])"
