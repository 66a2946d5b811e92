#!/usr/bin/env python
"""cd_life_kernel (round 5) against the round-4 lifecycle kernel over problem sizes, same runs: kernel time of K steps of R restarts.
usage: python tools/life_vs_r4.py [K=20] [R=4096]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcqp_amd import problems  # noqa: E402
from qcqp_amd.engine import Engine  # noqa: E402
from qcqp_amd.form import QCQPForm  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
for n in (64, 128, 256, 384, 512, 640, 768, 896, 1024):
    funcs, _, _ = problems.boolean_least_squares(n, n // 4, seed=1)
    form = QCQPForm.from_arrays(funcs)
    row = []
    for ver in (2, 1):
        e = Engine(form)
        e.cd_life_version(ver)
        e.cd_stream_run(K, R, seed=5, seed_stride=1)
        ms = []
        for rep in range(3):
            o = e.cd_stream_run(K, R, seed=100 + rep, seed_stride=1)
            ms.append(e.kernel_ms(2))
        sw = float(o['visits2'].sum()) / n
        row.append((min(ms), sw * 2.0 * n * n / (min(ms) * 1e-3) / 78.6e12, e.last_cd_kernel()))
        e.close()
    fac = None
    if n >= 128:       # round 6: the factored instantiation (P0 = L L^T of rank n / 4)
        from qcqp_amd import lowrank
        P0 = funcs[0][0]
        Lf = lowrank.objective_factor(P0.toarray() if hasattr(P0, 'toarray') else np.asarray(P0), max_rank=288)
        if Lf is not None:
            e = Engine(form)
            e.cd_set_objective_factor(Lf)
            e.cd_stream_run(K, R, seed=5, seed_stride=1)
            ms = []
            for rep in range(3):
                o = e.cd_stream_run(K, R, seed=100 + rep, seed_stride=1)
                ms.append(e.kernel_ms(2))
            sw = float(o['visits2'].sum()) / n
            fac = (min(ms), sw * 2.0 * n * n / (min(ms) * 1e-3) / 78.6e12, e.last_cd_kernel())
            e.close()
    print('n = %4d: %s %.3f ms (frac %.3f) | %s %.3f ms (frac %.3f) | round 5 / round 4 time %.2f | factored: %s' % (
        n, row[0][2], row[0][0], row[0][1], row[1][2], row[1][0], row[1][1], row[0][0] / row[1][0],
        'n/a' if fac is None else '%s %.3f ms (frac %.3f), %.2f of the faster of the other two' % (fac[2], fac[0], fac[1], fac[0] / min(row[0][0], row[1][0]))), flush=True)
