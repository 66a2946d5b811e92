mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_gpu_life.py -q -m gpu --durations=16 2>&1 | tail -60 > gpurun_out/t_life.log; tail -60 gpurun_out/t_life.log
