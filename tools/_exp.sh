cd $GRAFT_REPO_ROOT
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(1e3*d['timed_region_s'],2), round(d['roofline']['kernel_ms_per_launch'],2))"; }
export OPENBLAS_NUM_THREADS=1 OMP_NUM_THREADS=1 MKL_NUM_THREADS=1
for i in 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16; do run blas1 ""; done
unset OPENBLAS_NUM_THREADS OMP_NUM_THREADS MKL_NUM_THREADS
for i in 1 2 3 4 5 6 7 8; do run nofactor "--no-factor"; done
