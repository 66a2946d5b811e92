mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_admm_sep.py -x -q -m gpu > gpurun_out/t_admm.log 2>&1; grep -E "passed|failed|Error|^E " gpurun_out/t_admm.log | cut -c1-220
timeout 1500 python -m pytest tests/test_gpu_scale.py tests/test_gpu_fullsize.py tests/test_gpu_api.py -x -q -m gpu -k "admm or ADMM or api" > gpurun_out/t_admm2.log 2>&1; tail -3 gpurun_out/t_admm2.log
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_admm_sep_r05; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python $GRAFT_REPO_ROOT/tools/admm_sep_rate.py 1024 4096 100 --gemm > $OUT/stats.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
grep -v "^[WE]2026" $OUT/stats.log | tail -5
head -10 $OUT/stats/stats_kernel_stats.csv | cut -c1-150
