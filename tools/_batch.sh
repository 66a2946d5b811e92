mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_life.py tests/test_gpu_stream.py -x -q -m gpu > gpurun_out/t_life.log 2>&1; tail -3 gpurun_out/t_life.log
timeout 600 python tools/life_vs_r4.py > gpurun_out/life_vs_r4.log 2>&1; tail -20 gpurun_out/life_vs_r4.log
