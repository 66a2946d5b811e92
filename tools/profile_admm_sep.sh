#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel statistics (+ VALU / busy counters of the unit step) of improve(ADMM) on
# Boolean least squares through unit bases (tools/admm_sep_rate.py 1024 4096 100).  Output: gpurun_out/prof_admm_sep_$TAG/.
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_admm_sep_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/admm_sep_rate.py 1024 4096 100"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/valu -o valu --output-format csv -- $CMD > $OUT/valu.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
grep -v "^[WE]2026" $OUT/stats.log | tail -4
head -12 $OUT/stats/stats_kernel_stats.csv | cut -c1-160
