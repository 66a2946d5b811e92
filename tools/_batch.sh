mkdir -p gpurun_out
{
for i in 1 2; do timeout 300 python tools/stream_check.py 1024 4096 20 1000 2>&1 | grep -E "stream run 2|profile|worst|kernels|rror"; done
} > gpurun_out/r4b.log 2>&1
cat gpurun_out/r4b.log
