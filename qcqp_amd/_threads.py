"""Host threads: how many cores this process may really use, and a limit for the BLAS pools of the host-side linear algebra.

Why (round 6, profiles/r06_timed_region.md): on the GPU boxes the container sees 256 CPUs but its cgroup grants 16 cores of CPU
time per 100 ms (cpu.max = "1600000 100000").  OpenBLAS starts a thread per visible CPU; after a product its workers spin for a
while, the 16-core budget of the period is spent in a few milliseconds and the kernel throttles EVERY thread of the container until
the period ends -- the host thread that waits for a 34 ms launch then returns 10-40 ms late (one bench run in four, always in the
call that followed the pivoted Cholesky of qcqp_amd.lowrank.objective_factor).  Limiting the pools to what the cgroup grants
removes the stalls (0 of 16 runs)."""
import contextlib
import os


def usable_cores():
    """min(affinity mask, cgroup CPU quota); at least 1."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max',):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != 'max':
                cores = min(cores, max(1, int(int(quota) // int(period))))
        except (OSError, ValueError):
            pass
    try:
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0 and p > 0:
            cores = min(cores, max(1, q // p))
    except (OSError, ValueError):
        pass
    return max(1, cores)


def local_ranks():
    """Processes of this job that share the machine (and its CPU quota): the launcher's LOCAL_WORLD_SIZE / WORLD_SIZE, or what
    bench.py announces for the ranks it spawns itself (QCQP_LOCAL_RANKS)."""
    n = 1
    for v in ('LOCAL_WORLD_SIZE', 'WORLD_SIZE', 'QCQP_LOCAL_RANKS'):
        try:
            n = max(n, int(os.environ.get(v, '1')))
        except ValueError:
            pass
    return n


def blas_threads():
    """Threads for the BLAS pools: half of this rank's share of what the job may use (the rest is for the thread that drives the
    GPU -- it polls -- and for the CPU-baseline workers), at most 8."""
    return max(1, min(8, usable_cores() // (2 * local_ranks())))


def set_blas_env():
    """Before NumPy is imported: cap the pools through the environment (does not override the user's own settings)."""
    n = str(blas_threads())
    for v in ('OPENBLAS_NUM_THREADS', 'OMP_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ.setdefault(v, n)


def blas_limit():
    """Context manager: cap the pools of an already imported NumPy / SciPy (threadpoolctl; a no-op without it)."""
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        return contextlib.nullcontext()
    return threadpool_limits(limits=blas_threads())
