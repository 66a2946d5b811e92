#!/bin/bash
# rocprofv3 kernel statistics of the dense-path scale check (run through gpurun); args: R iters n m
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_dense
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o d_$1 --output-format csv -- python $R/tools/scale_checks.py cfg5 $1 $2 $3 $4 > $OUT/log_$1.txt 2>&1
grep -E "cfg5|phase|cd_run" $OUT/log_$1.txt
head -8 $OUT/d_$1_kernel_stats.csv | cut -c1-150
