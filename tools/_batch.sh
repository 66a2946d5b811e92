mkdir -p gpurun_out
(
echo "=== bls 1024 4096 K=20"; LIFE_PROF=1 LIFE_SERIAL=0 timeout 300 python tools/life_check.py bls 1024 4096 20 1000 0
) > gpurun_out/lc14.log 2>&1
grep -v "^$" gpurun_out/lc14.log | cut -c1-330 | grep -v "run [01]:" | tail -12
timeout 2400 python -m pytest tests -q -m gpu --durations=10 -x 2>&1 | tail -30 > gpurun_out/t_all.log; tail -30 gpurun_out/t_all.log
