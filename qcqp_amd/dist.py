"""Multi-GPU sharding: one process per GPU, restarts/samples partitioned by GLOBAL index.

There is no data-path collective: every rank holds a replica of the problem and runs its own
slice of the restarts; the only exchange is the final "pick the global best" step
(QCQPForm.better ordering, utilities.py:135-146), done natively by RCCL inside the C library
(qcqpmi_comm_select_best).  The RCCL unique id is the only thing that has to travel between the
processes beforehand; that bootstrap uses torch.distributed's gloo store when the job was
launched by torch.distributed.run (plumbing only -- no tensor of the hot path touches torch).
"""
import os

import numpy as np


def env_world():
    """(rank, local_rank, world) from the launcher's environment (torch.distributed.run)."""
    return (int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)),
            int(os.environ.get('WORLD_SIZE', 1)))


def shard_range(total, rank, world):
    """Contiguous slice [first, first+count) of `total` global indices owned by `rank`."""
    base, rem = divmod(int(total), int(world))
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def better_key(f0, maxviol, index, tol=1e-4):
    """Total order equivalent to folding QCQPForm.better over candidates, ties -> lowest index."""
    v = float(maxviol)
    bucket = int(v / tol) if v == v else 1 << 62
    return (bucket, float(f0), int(index))


def select_best_host(f0, maxviol, tol=1e-4, index_offset=0):
    """Reference implementation of the selection rule on host arrays (used by tests and by the
    gloo transport of the CPU-only multi-process tests)."""
    best = None
    for i, (f, v) in enumerate(zip(f0, maxviol)):
        k = better_key(f, v, index_offset + i, tol)
        if best is None or k < best:
            best = k
    return best


class GlooBootstrap(object):
    """torch.distributed (gloo) used ONLY to move small Python objects between ranks."""

    def __init__(self):
        import torch.distributed as td
        self.td = td
        if not td.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            td.init_process_group(backend='gloo')
        self.rank, self.world = td.get_rank(), td.get_world_size()

    def broadcast_bytes(self, payload, src=0):
        obj = [payload if self.rank == src else None]
        self.td.broadcast_object_list(obj, src=src)
        return obj[0]

    def allgather(self, obj):
        out = [None] * self.world
        self.td.all_gather_object(out, obj)
        return out

    def barrier(self):
        self.td.barrier()


def init_rccl(engine, rank, world, bootstrap=None):
    """Create the RCCL communicator of `engine` (native, inside libqcqp_mi.so)."""
    if world == 1:
        uid = engine.comm_unique_id()
    else:
        bootstrap = bootstrap or GlooBootstrap()
        uid = engine.comm_unique_id().tobytes() if rank == 0 else None
        uid = np.frombuffer(bootstrap.broadcast_bytes(uid, 0), dtype=np.uint8).copy()
    engine.comm_init(rank, world, uid)
    return bootstrap


def global_best_gloo(bootstrap, local_key, local_x):
    """CPU transport of the final exchange (tests only): all-gather the keys, take the minimum,
    winner's x is broadcast.  Mirrors qcqpmi_comm_select_best."""
    keys = bootstrap.allgather(local_key)
    win = min(range(len(keys)), key=lambda w: keys[w])
    x = bootstrap.broadcast_bytes(np.asarray(local_x, dtype=np.float64).tobytes()
                                  if bootstrap.rank == win else None, win)
    return keys[win], np.frombuffer(x, dtype=np.float64).copy()
