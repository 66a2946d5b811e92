#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel statistics of the default bench command,
# then the two HBM PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots).
# Output goes to gpurun_out/prof_$TAG/; tools/summarize_profile.py turns it into profiles/.
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the driver's step count; warm-up of the same size, so that every launch of the lifecycle kernel in the trace is 20 steps
CMD="python $ROOT/bench.py --scheme stream --steps 20 --warmup 20 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o fetch --output-format csv -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o write --output-format csv -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/l2 -o l2 --output-format csv -- $CMD > $OUT/l2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/mfma -o mfma --output-format csv -- $CMD > $OUT/mfma.log 2>&1
grep '^{' $OUT/stats.log | tail -1 > $OUT/bench_under_profiler.json
ls -R $OUT | head -40
