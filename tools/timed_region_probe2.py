#!/usr/bin/env python
"""First K-step call after a W-step warm-up in a fresh context, many times: how often is wall >> kernel?  (round 6: the driver's
bench line showed 73 ms of timed region around a 34 ms launch in about one run of four)
usage: python tools/timed_region_probe2.py mode [trials]     mode: warm5 | warm20 | reserve_first"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcqp_amd import lowrank, problems  # noqa: E402
from qcqp_amd.engine import Engine  # noqa: E402
from qcqp_amd.form import QCQPForm  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'warm5'
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 12
K, R, n = 20, 4096, 1024
funcs = problems.boolean_least_squares(n, n // 4, seed=1)[0]
form = QCQPForm.from_arrays(funcs)
P0 = np.asarray(funcs[0][0].toarray() if hasattr(funcs[0][0], 'toarray') else funcs[0][0])
Lf = lowrank.objective_factor(P0)
out = []
for t in range(trials):
    eng = Engine(form)
    eng.cd_set_objective_factor(Lf)
    if mode == 'reserve_first':
        eng.cd_stream_reserve(K, R)
    eng.cd_stream_run(K if mode == 'warm20' else 5, R, seed=1, seed_stride=1)
    eng.cd_stream_reserve(K, R)
    eng.sync()
    t0 = time.perf_counter()
    eng.cd_stream_run(K, R, seed=100 + t, seed_stride=1)
    t1 = time.perf_counter()
    out.append((1e3 * (t1 - t0), eng.kernel_ms(2)))
    del eng
print(mode, ' '.join('%.1f/%.1f' % w for w in out))
print(mode, 'outliers (wall > kernel + 3 ms): %d of %d' % (sum(1 for w, k in out if w > k + 3.0), len(out)))
