#!/usr/bin/env python
"""Lifecycle (population-streaming) run against K serial suggest(RANDOM) + improve(COORD_DESCENT) calls: points, counters,
objective, best restart; and its rate.  Usage: python tools/stream_check.py [n] [R] [K] [num_iters]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from qcqp_amd import problems  # noqa: E402
from qcqp_amd.engine import Engine  # noqa: E402
from qcqp_amd.form import QCQPForm  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
    cs = int(sys.argv[5]) if len(sys.argv) > 5 else -1      # experiments: blocks of the contraction the chain wave multiplies (default 4)
    funcs, _, _ = problems.boolean_least_squares(n, max(4, n // 4), seed=1)
    form = QCQPForm.from_arrays(funcs)
    e = Engine(form)
    seed0, first0, fstride = 1000, 7, 100000
    ref = []
    for p in range(K):
        e.randn(R, seed=seed0 + p, first_index=first0 + p * fstride)
        out = e.cd_run(phase1=True, num_iters=iters, seed=seed0 + p, first_index=first0 + p * fstride)
        ref.append((e.download(), out, e.select_best(), e.last_cd_kernel()))
    es = Engine(form)
    if cs >= 0:
        es.L.qcqpmi_debug_profile(es.h, (128 | (cs << 8)) << 4, None)
    for rep in range(3):
        t0 = time.perf_counter()
        o = es.cd_stream_run(K, R, num_iters=iters, seed=seed0, seed_stride=1, first_index=first0, first_stride=fstride)
        dt = time.perf_counter() - t0
        ms = es.kernel_ms(2)
        sweeps = float(o['visits2'].sum()) / n
        print('stream run %d: wall %.2f ms, kernel %.3f ms, %.3e restart-sweeps/s (kernel), frac %.3f; per population %.3f ms' % (
            rep, 1e3 * dt, ms, sweeps / (ms * 1e-3), sweeps * 2.0 * n * n / (ms * 1e-3) / 78.6e12, ms / K))
    import ctypes as C
    es.L.qcqpmi_debug_profile(es.h, 1 | (((128 | (cs << 8)) << 4) if cs >= 0 else 0), None)
    o2 = es.cd_stream_run(K, R, num_iters=iters, seed=seed0, seed_stride=1, first_index=first0, first_stride=fstride)
    pr = np.zeros(24, dtype=np.int64)
    es.L.qcqpmi_debug_life_profile(es.h, pr.ctypes.data_as(C.POINTER(C.c_int64)))
    es.L.qcqpmi_debug_profile(es.h, ((128 | (cs << 8)) << 4) if cs >= 0 else 0, None)
    print('profile: column build %.1f %% of the workgroups\' time (normals %.1f %%), roles %.1f %%, write-out / queue / idle at the end %.1f %%; %d episodes, %d columns' % (
        100.0 * pr[0] / max(pr[1], 1), 100.0 * pr[4] / max(pr[1], 1), 100.0 * pr[5] / max(pr[1], 1), 100.0 * (pr[1] - pr[0] - pr[5]) / max(pr[1], 1), pr[2], pr[3]))
    X = es.download()
    print('kernels:', ref[0][3], '/', es.last_cd_kernel())
    worst = 0.0
    for p in range(K):
        Xr, outr, best, _ = ref[p]
        sl = slice(p * R, (p + 1) * R)
        d = np.max(np.abs(X[:, sl] - Xr), axis=0)
        worst = max(worst, d.max())
        same = {k: bool(np.array_equal(o[k][sl], outr[k])) for k in ('sweeps1', 'sweeps2', 'visits2', 'accepted2', 'ran_phase2', 'status1', 'status2')}
        rf = np.max(np.abs(o['f0'][sl] - outr['f0']) / (1 + np.abs(outr['f0'])))
        rv = np.max(np.abs(o['maxviol'][sl] - outr['maxviol']))
        print('population %d: max |dx| %.2e (restarts off: %d), counters equal: %s, rel df0 %.2e, d maxviol %.2e; best %d (serial %d), f0 %.12g (%.12g)' % (
            p, d.max(), int((d > 1e-9).sum()), same, rf, rv, o['best_index'][p], best[0], o['best_f0'][p], best[1]))
        assert o['best_index'][p] == best[0], (o['best_index'][p], best[0])
        assert np.max(np.abs(o['best_x'][p] - best[3])) <= 1e-9
    print('worst |dx|', worst)


if __name__ == '__main__':
    main()
