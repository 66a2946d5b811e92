#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel statistics + PMC passes of the ADMM improve path at BASELINE.json
# configs[3] size (tools/admm_fused_rate.py: reduced bases, 1024 restarts, num_iters = 1000 -- the fused persistent kernel,
# then the multi-launch path, then the fused kernel again).  Output: gpurun_out/prof_admm_$TAG/.
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_admm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/admm_fused_rate.py 1024 1000"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $CMD > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/mfma -o mfma --output-format csv -- $CMD > $OUT/mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o fetch --output-format csv -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o write --output-format csv -- $CMD > $OUT/write.log 2>&1
python $ROOT/tools/admm_fused_rate.py 1024 1000 --prof > $OUT/stages.log 2>&1
tail -4 $OUT/stats.log; tail -3 $OUT/stages.log
ls -R $OUT | head -30
