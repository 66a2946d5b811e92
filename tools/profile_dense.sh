#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel statistics + PMC passes of the dense-constraint coordinate descent
# (BASELINE.json configs[4] family at n = 1024, m = 256, 512 and 4096 restarts: tools/dense_rate.py).  Output: gpurun_out/prof_dense_$TAG/.
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_dense_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/dense_rate.py 1024 256 512 4096"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $CMD > $OUT/stats.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/mfma -o mfma --output-format csv -- $CMD > $OUT/mfma.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o fetch --output-format csv -- $CMD > $OUT/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o write --output-format csv -- $CMD > $OUT/write.log 2>&1
grep "^n " $OUT/stats.log
find $OUT -name "*kernel_trace.csv" -delete < /dev/null
ls -R $OUT < /dev/null | head -30
