#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel statistics of the ADMM improve path at BASELINE.json configs[3]
# size (tools/admm_scale.py: reduced bases, 1024 restarts, 40 iterations per phase).  Output: gpurun_out/prof_admm_$TAG/.
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_admm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python $ROOT/tools/admm_scale.py 512 1024 40 > $OUT/stats.log 2>&1
tail -3 $OUT/stats.log
