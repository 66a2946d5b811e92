mkdir -p gpurun_out
(
echo "=== bls 1024 4096 K=20"; LIFE_PROF=1 LIFE_SERIAL=0 timeout 300 python tools/life_check.py bls 1024 4096 20 1000 0
echo "=== bls 1024 4096 K=20 256 wgs"; LIFE_DBG=$((1024 | (256<<12))) LIFE_PROF=1 LIFE_SERIAL=0 timeout 300 python tools/life_check.py bls 1024 4096 20 1000 0
) > gpurun_out/lc7.log 2>&1
grep -v "^$" gpurun_out/lc7.log | cut -c1-330 | grep -v "^bls.*run [01]" | tail -70
