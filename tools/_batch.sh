mkdir -p gpurun_out
(
echo "=== bls 1024 4096 K=20"; LIFE_PROF=1 LIFE_SERIAL=0 timeout 300 python tools/life_check.py bls 1024 4096 20 1000 0
echo "=== bls 1000"; timeout 200 python tools/life_check.py bls 1000 600 2 1000 0
echo "=== box 100"; timeout 120 python tools/life_check.py box 100 60 2 1000 0
) > gpurun_out/lc12.log 2>&1
grep -v "^$" gpurun_out/lc12.log | cut -c1-330 | grep -v "run [01]:" | tail -30
