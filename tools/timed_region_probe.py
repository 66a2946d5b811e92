#!/usr/bin/env python
"""Where the wall clock of one headline step batch goes: qcqpmi_cd_stream_run (launch + fetch + per-population best) against the
kernel's HIP-event time, with and without the objective factor.  Usage: python tools/timed_region_probe.py [K] [R] [n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcqp_amd import lowrank, problems  # noqa: E402
from qcqp_amd.engine import Engine  # noqa: E402
from qcqp_amd.form import QCQPForm  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
funcs = problems.boolean_least_squares(n, n // 4, seed=1)[0]
form = QCQPForm.from_arrays(funcs)
for factor in ((True,) if os.environ.get('PROBE_FACTOR_ONLY') else (False, True)):
    eng = Engine(form)
    if factor:
        P0 = np.asarray(funcs[0][0].toarray() if hasattr(funcs[0][0], 'toarray') else funcs[0][0])
        eng.cd_set_objective_factor(lowrank.objective_factor(P0))
    eng.cd_stream_run(5, R, seed=1, seed_stride=1)
    eng.cd_stream_reserve(K, R)
    eng.sync()
    for rep in range(int(os.environ.get('PROBE_REPS', '4'))):
        t0 = time.perf_counter()
        o = eng.cd_stream_run(K, R, seed=100 + rep, seed_stride=1, want_best_x=(rep % 2 == 0))
        t1 = time.perf_counter()
        print('factor %d rep %d (best_x %d): wall %.2f ms, kernel %.2f ms (%s)' % (factor, rep, rep % 2 == 0, 1e3 * (t1 - t0), eng.kernel_ms(2), eng.last_cd_kernel()), flush=True)
