cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_dense_r06; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python $GRAFT_REPO_ROOT/tools/dense_rate.py 1024 256 512 > $OUT/stats.log 2>&1
grep -v "^[WE]2026" $OUT/stats.log | tail -3
head -12 $OUT/stats/stats_kernel_stats.csv | cut -c1-140
python $GRAFT_REPO_ROOT/tools/dense_timeline.py $OUT/stats 40
find $OUT -name "*kernel_trace.csv" -size +20M -delete
