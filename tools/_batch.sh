mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_admm_sep.py tests/test_gpu_api.py -x -q -m gpu -s > gpurun_out/t_admm.log 2>&1; grep -E "ADMM unit|f0 of the|passed|failed|Error|^E " gpurun_out/t_admm.log | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "admm" > gpurun_out/t_admm2.log 2>&1; tail -3 gpurun_out/t_admm2.log
BENCH_ONLY=1 timeout 600 python tools/_dbg_bench.py 2>&1 | grep -E '"value"|wall_s|frac"'
