"""Multi-GPU sharding: one process per GPU, restarts/samples partitioned by GLOBAL index.

There is no data-path collective: every rank holds a replica of the problem and runs its own
slice of the restarts; the only exchange is the final "pick the global best" step
(QCQPForm.better ordering, utilities.py:135-146), done natively by RCCL inside the C library
(qcqpmi_comm_select_best).  The 128-byte RCCL unique id is the only thing that has to travel
between the processes beforehand.  That bootstrap is a tiny file rendezvous in a per-job
directory on the node's local file system (one node = one file system; no torch, no sockets):
every message is a file written atomically (tmp + rename), readers poll for it.

Two ways to get N ranks:
  * a launcher sets RANK / LOCAL_RANK / WORLD_SIZE (`python -m torch.distributed.run ...` does) and
    every process calls `env_world()`;
  * `spawn_local_ranks(N, argv)` from a plain `python bench.py --gpus N`: the calling process becomes
    rank 0 and N-1 children are started with the same command line.
"""
import json
import os
import stat
import subprocess
import sys
import tempfile
import time
import warnings
import uuid

import numpy as np

RDZV_ENV = 'QCQP_AMD_RDZV'


def env_world():
    """(rank, local_rank, world) from the launcher's environment."""
    return (int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', os.environ.get('RANK', 0))),
            int(os.environ.get('WORLD_SIZE', 1)))


def shard_range(total, rank, world):
    """Contiguous slice [first, first+count) of `total` global indices owned by `rank`."""
    base, rem = divmod(int(total), int(world))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def better_key(f0, maxviol, index, tol=1e-4):
    """Total order equivalent to folding QCQPForm.better over candidates, ties -> lowest index."""
    v = float(maxviol)
    bucket = int(v / tol) if (v == v and v != float('inf')) else 1 << 62
    return (bucket, float(f0), int(index))


def select_best_host(f0, maxviol, tol=1e-4, index_offset=0):
    """Reference implementation of the selection rule on host arrays (used by tests and by the
    file transport of the CPU-only multi-process tests)."""
    best = None
    for i, (f, v) in enumerate(zip(f0, maxviol)):
        k = better_key(f, v, index_offset + i, tol)
        if best is None or k < best:
            best = k
    return best


def global_best_of_populations(allreduce_sum, rank, world, f0, maxviol, gindex, xs, tol=1e-4, allgather=None):
    """The global best of K populations at once -- ONE exchange for a whole streamed run instead of one per population.
    Every rank holds, per population k, its local winner (f0[k], maxviol[k], gindex[k] = GLOBAL restart index, xs[k] = the
    point).  Round 1, the keys: `allgather(array) -> (world,) + array.shape` (Engine.comm_allgather: one ncclAllGather) of the
    K records (bucket:int64, f0:fp64 bits, maxviol:fp64 bits, gindex:int64) = 32 bytes per population and rank -- the
    QCQPForm.better ordering (utilities.py:135-146) on (bucket, f0) with the global index as the final tie-break; integers
    travel as integers.  Everyone then knows the owner of every population.  Round 2, the points: `allreduce_sum(array) -> array`
    (Engine.comm_allreduce: one all-reduce) of the (K, n) table with only the rows this rank owns filled.  Bytes a rank sends:
    32 K for the keys + at most 2 x 8 K n for the points (ring all-reduce), independent of the number of ranks.
    Without `allgather` (older callers) the keys travel through a sum-all-reduce of the (world, K, 3) table like in round 4.
    Returns ([(gindex, f0, maxviol)] * K, X (K, n)), identical on every rank."""
    f0 = np.asarray(f0, dtype=np.float64)
    K = f0.size
    xs = np.asarray(xs, dtype=np.float64).reshape(K, -1)
    mv = np.asarray(maxviol, dtype=np.float64)
    gi = np.asarray(gindex, dtype=np.int64)
    if world > 1 and allgather is not None:
        rec = np.zeros((K, 4), dtype=np.int64)
        rec[:, 0] = [better_key(f0[k], mv[k], 0, tol)[0] for k in range(K)]
        rec[:, 1] = f0.view(np.int64)
        rec[:, 2] = mv.view(np.int64)
        rec[:, 3] = gi
        allrec = np.asarray(allgather(rec)).reshape(world, K, 4)
        fall = np.ascontiguousarray(allrec[:, :, 1]).view(np.float64)
        vall = np.ascontiguousarray(allrec[:, :, 2]).view(np.float64)
        iall = allrec[:, :, 3]
        owner = [min(range(world), key=lambda g: (int(allrec[g, k, 0]), float(fall[g, k]), int(iall[g, k]))) for k in range(K)]
    else:
        keys = np.zeros((world, K, 3))
        keys[rank, :, 0], keys[rank, :, 1], keys[rank, :, 2] = f0, mv, gi.astype(np.float64)
        if world > 1:
            keys = np.asarray(allreduce_sum(keys.ravel())).reshape(world, K, 3)
        fall, vall, iall = keys[:, :, 0], keys[:, :, 1], keys[:, :, 2].astype(np.int64)
        owner = [min(range(world), key=lambda g: better_key(fall[g, k], vall[g, k], int(iall[g, k]), tol)) for k in range(K)]
    X = np.zeros_like(xs)
    for k in range(K):
        if owner[k] == rank:
            X[k] = xs[k]
    if world > 1:
        X = np.asarray(allreduce_sum(X.ravel())).reshape(K, -1)
    return [(int(iall[owner[k], k]), float(fall[owner[k], k]), float(vall[owner[k], k])) for k in range(K)], X


def _job_key():
    """Directory name every rank of one job agrees on without talking to each other."""
    key = os.environ.get(RDZV_ENV)
    if key:
        return key
    # launched by an external launcher: all workers share the launcher's pid and rendezvous endpoint
    return 'ext_%s_%s_%s_%d' % (os.environ.get('MASTER_ADDR', 'local'), os.environ.get('MASTER_PORT', '0'),
                                os.environ.get('TORCHELASTIC_RUN_ID', 'none'), os.getppid())


RANK_START_WINDOW = 300.0      # seconds within which all ranks of one job start


def _job_epoch():
    """Earliest time a message of THIS job can have been written (external launcher: the key is predictable, so files of an
    earlier job with the same key must not be read as messages): the later of
      * the start of the launcher (the parent process) -- a crashed job with a recycled launcher pid is older than that;
      * this rank's own start minus RANK_START_WINDOW -- ranks started one after the other by the SAME parent (a shell script:
        same ppid, same MASTER_PORT, hence the same key and the same launcher start) still reject what a job that crashed more
        than the window ago left behind.  Inside the window the files of a crashed twin job cannot be told from this job's
        by time alone: rank 0 therefore also clears what is older than the epoch when it opens the directory (FileComm)."""
    if os.environ.get(RDZV_ENV):
        return 0.0                      # uuid key: the directory cannot pre-exist
    launcher = _proc_start_time(os.getppid())
    own = _proc_start_time(os.getpid())
    if own <= 0.0:
        own = time.time()
    if launcher <= 0.0:
        warnings.warn('qcqp_amd.dist: the start time of the launcher (pid %d) is unavailable; messages older than %.0f s before this '
                      'rank started are treated as stale' % (os.getppid(), RANK_START_WINDOW))
        return own - RANK_START_WINDOW
    return max(launcher - 2.0, own - RANK_START_WINDOW)


def _proc_start_time(pid):
    """Start time of a process as seconds since the epoch: field 22 of /proc/<pid>/stat (clock ticks since boot) + btime of
    /proc/stat.  (The mtime of /proc/<pid> is NOT the start time: Linux stamps the inode when it is first looked up, so ranks
    that start at different moments would compute different epochs and reject each other's files.)  0.0 if unavailable."""
    try:
        with open('/proc/%d/stat' % pid) as f:
            fields = f.read().rsplit(')', 1)[1].split()      # the command name may contain spaces and parentheses
        ticks = float(fields[19])                            # field 22 overall = index 19 after "pid (comm)"
        btime = None
        with open('/proc/stat') as f:
            for line in f:
                if line.startswith('btime'):
                    btime = float(line.split()[1])
                    break
        if btime is None:
            return 0.0
        return btime + ticks / float(os.sysconf('SC_CLK_TCK'))
    except (OSError, ValueError, IndexError):
        return 0.0


def _private_dir(path):
    """mkdir -p with mode 0700, then insist that the directory is ours and closed to group / others (a shared /tmp:
    another local user must not be able to pre-create the job directory and plant messages)."""
    try:
        os.makedirs(path, mode=0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError('rendezvous directory %s is not a private directory of uid %d (mode %o, owner %d)'
                           % (path, os.getuid(), st.st_mode & 0o7777, st.st_uid))
    return path


def _rdzv_root():
    base = os.environ.get('XDG_RUNTIME_DIR')
    if base and os.path.isdir(base) and os.access(base, os.W_OK):
        return _private_dir(os.path.join(base, 'qcqp_amd_rdzv'))
    return _private_dir(os.path.join(tempfile.gettempdir(), 'qcqp_amd_rdzv_%d' % os.getuid()))


def _encode(obj):
    """Messages are raw bytes (the 128-byte RCCL id, a point) or small JSON values (keys, None): nothing executable
    is ever deserialised (round 2 used pickle)."""
    if isinstance(obj, (bytes, bytearray, memoryview)):
        return b'B' + bytes(obj)
    return b'J' + json.dumps(obj).encode('ascii')


def _tuplify(v):
    return tuple(_tuplify(a) for a in v) if isinstance(v, list) else v


def _decode(raw):
    if raw[:1] == b'B':
        return raw[1:]
    if raw[:1] == b'J':
        return _tuplify(json.loads(raw[1:].decode('ascii')))
    raise RuntimeError('malformed rendezvous message')


class FileRendezvous(object):
    """Moves small messages (bytes / JSON values) between the ranks of one node through atomically renamed files in a
    directory only this user can enter."""

    def __init__(self, rank=None, world=None, key=None, timeout=600.0):
        r, _, w = env_world()
        self.rank = r if rank is None else int(rank)
        self.world = w if world is None else int(world)
        self.timeout = float(timeout)
        key = key or _job_key()
        if os.sep in key or key in ('.', '..'):
            raise ValueError('bad rendezvous key %r' % key)
        self.dir = _private_dir(os.path.join(_rdzv_root(), key))
        self.epoch = _job_epoch()
        self.seq = 0
        # leftovers of a dead job with the same key: rank 0 removes what NO rank of this job can have written, i.e. what is older
        # than the launcher all ranks share -- not what is older than rank 0's own epoch: a rank 0 that starts more than
        # RANK_START_WINDOW after another rank (sequential shell launch, slow scheduler) would delete that rank's live messages
        # and the job would hang instead of running.  (Such a late rank 0 still REJECTS those messages as stale: the reader's
        # timeout below then names the window.)
        launcher = 0.0 if os.environ.get(RDZV_ENV) else _proc_start_time(os.getppid())
        if self.rank == 0 and launcher > 0.0:
            for name in os.listdir(self.dir):
                path = os.path.join(self.dir, name)
                try:
                    if os.lstat(path).st_mtime < launcher - 2.0:
                        os.remove(path)
                except OSError:
                    pass

    def _put(self, name, obj):
        path = os.path.join(self.dir, name)
        tmp = '%s.tmp.%d' % (path, os.getpid())
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
        with os.fdopen(fd, 'wb') as f:
            f.write(_encode(obj))
        os.rename(tmp, path)

    def _fresh(self, path):
        try:
            st = os.lstat(path)
        except OSError:
            return False
        if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid():
            raise RuntimeError('foreign file %s in the rendezvous directory' % path)
        return st.st_mtime >= self.epoch      # older: left behind by a dead job with the same key -> not a message

    def _get(self, name):
        path = os.path.join(self.dir, name)
        t0 = time.time()
        delay = 0.0005
        while not self._fresh(path):
            if time.time() - t0 > self.timeout:
                stale = os.path.exists(path)
                raise RuntimeError('rendezvous timeout waiting for %s (rank %d of %d)%s' % (path, self.rank, self.world,
                                   '; the file exists but is older than this rank accepts: all ranks of a job must start within '
                                   '%.0f s of each other (RANK_START_WINDOW), or set %s to a per-job key' % (RANK_START_WINDOW, RDZV_ENV)
                                   if stale else ''))
            time.sleep(delay)
            delay = min(delay * 1.5, 0.05)
        with open(path, 'rb') as f:
            return _decode(f.read())

    def broadcast_bytes(self, payload, src=0):
        self.seq += 1
        name = 'b%06d' % self.seq
        if self.rank == src:
            self._put(name, payload)
            return payload
        return self._get(name)

    def allgather(self, obj):
        self.seq += 1
        self._put('g%06d.%d' % (self.seq, self.rank), obj)
        return [self._get('g%06d.%d' % (self.seq, r)) for r in range(self.world)]

    def barrier(self):
        self.allgather(None)

    def close(self):
        """Last collective of a job: everybody arrives, then rank 0 removes the directory."""
        try:
            self.barrier()
            if self.rank == 0:
                time.sleep(0.05)
                for name in os.listdir(self.dir):
                    try:
                        os.remove(os.path.join(self.dir, name))
                    except OSError:
                        pass
                os.rmdir(self.dir)
        except Exception:
            pass


def spawn_local_ranks(world, argv=None, quiet=True):
    """Plain `python script.py --gpus N` (no launcher): start ranks 1..N-1 as children running the same
    command line, make the caller rank 0.  Returns the list of child processes (empty when a launcher
    already provided the world, or world == 1)."""
    world = int(world)
    if world <= 1 or 'WORLD_SIZE' in os.environ:
        return []
    key = 'job_%s' % uuid.uuid4().hex
    os.environ[RDZV_ENV] = key
    os.environ['WORLD_SIZE'] = str(world)
    os.environ['RANK'] = '0'
    os.environ['LOCAL_RANK'] = '0'
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    argv = list(sys.argv if argv is None else argv)
    kids = []
    for r in range(1, world):
        env = dict(os.environ)
        env['RANK'] = str(r)
        env['LOCAL_RANK'] = str(r)
        kids.append(subprocess.Popen([sys.executable] + argv, env=env,
                                     stdout=subprocess.DEVNULL if quiet else None))
    return kids


def wait_children(kids, timeout=600.0):
    """Reap the children of spawn_local_ranks; returns the worst exit code."""
    worst = 0
    t0 = time.time()
    for p in kids:
        try:
            rc = p.wait(timeout=max(1.0, timeout - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            p.kill()
            rc = -9
        worst = worst or rc
    return worst


def init_rccl(engine, rank, world, bootstrap=None):
    """Create the RCCL communicator of `engine` (native, inside libqcqp_mi.so).  Returns the
    rendezvous object (None for a single rank)."""
    if world == 1:
        uid = engine.comm_unique_id()
    else:
        bootstrap = bootstrap or FileRendezvous(rank, world)
        uid = engine.comm_unique_id().tobytes() if rank == 0 else None
        uid = np.frombuffer(bootstrap.broadcast_bytes(uid, 0), dtype=np.uint8).copy()
    engine.comm_init(rank, world, uid)
    return bootstrap


def init_file_comm(engine, rank, world, bootstrap=None):
    """The collectives bench.py needs -- barrier, all-reduce (max / sum) of float64 vectors, all-gather of raw arrays -- over the
    job's file rendezvous instead of RCCL, attached to `engine` under the names of its RCCL methods.  For ranks that SHARE a device
    (RCCL refuses a communicator with duplicate devices): two real engines in two processes on the one GPU of a test box exchange
    exactly what two GPUs would (tests/test_gpu_life.py).  Float64 values travel as their IEEE bytes: results are the bits RCCL's
    sum / max over two ranks would give (a + b is commutative; more than two ranks: summed in rank order)."""
    boot = bootstrap or (FileRendezvous(rank, world) if world > 1 else None)
    engine._world = int(world)

    def allgather(arr):
        a = np.ascontiguousarray(arr)
        if world == 1:
            return a.reshape((1,) + a.shape).copy()
        parts = boot.allgather(a.tobytes())
        return np.stack([np.frombuffer(p_, dtype=a.dtype).reshape(a.shape) for p_ in parts])

    def allreduce(values, op='max'):
        v = np.ascontiguousarray(np.atleast_1d(np.asarray(values, dtype=np.float64)))
        parts = allgather(v)
        out = parts[0].copy()
        for k in range(1, parts.shape[0]):
            out = np.maximum(out, parts[k]) if op == 'max' else out + parts[k]
        v[...] = out
        return v

    def barrier():
        if world > 1:
            boot.barrier()
    engine.comm_allgather, engine.comm_allreduce, engine.comm_barrier = allgather, allreduce, barrier
    return boot


def global_best_host(bootstrap, local_key, local_x):
    """CPU transport of the final exchange (tests only): all-gather the keys, take the minimum,
    winner's x is broadcast.  Mirrors qcqpmi_comm_select_best."""
    keys = bootstrap.allgather(tuple(local_key))
    win = min(range(len(keys)), key=lambda w: keys[w])
    x = bootstrap.broadcast_bytes(np.asarray(local_x, dtype=np.float64).tobytes()
                                  if bootstrap.rank == win else None, win)
    return keys[win], np.frombuffer(x, dtype=np.float64).copy()
