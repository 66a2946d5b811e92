mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_scale.py -q -m gpu -x -k "follows_the_oracle" -s 2>&1 | grep -v "^$" | tail -30 > gpurun_out/t_dense.log; tail -30 gpurun_out/t_dense.log
