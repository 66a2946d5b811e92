#!/usr/bin/env python
"""improve(ADMM) on Boolean least squares (BASELINE.json configs[1]'s problem) through bases of unit vectors: rate of the run.
usage: python tools/admm_sep_rate.py [n=1024] [R=4096] [iters=100] [--gemm]   (--gemm: also the GEMM path on the same bases)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcqp_amd import problems  # noqa: E402
from qcqp_amd.engine import Engine  # noqa: E402
from qcqp_amd.form import QCQPForm  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith('--')]
n = int(args[0]) if len(args) > 0 else 1024
R = int(args[1]) if len(args) > 1 else 4096
iters = int(args[2]) if len(args) > 2 else 100
funcs, _, _ = problems.boolean_least_squares(n, n // 4, seed=1)
form = QCQPForm.from_arrays(funcs)
e = Engine(form)
t0 = time.perf_counter()
e.admm_set_basis(*form.unit_bases())
rho = 50.0 / form.m
res, its = e.admm_zsolver_device(rho)
e.sync()
print('n=%d m=%d R=%d num_iters=%d rho=%.4g; setup %.3f s (Newton-Schulz: %d iterations, residual %.1e)' % (n, form.m, R, iters, rho, time.perf_counter() - t0, its, res))
for unit in ((True, False) if '--gemm' in sys.argv else (True,)):
    e.admm_unit_bases(unit)
    e.randn(R, seed=3)
    e.admm_run(rho, None, phase1=True, num_iters=2)
    for rep in range(2):
        e.randn(R, seed=3)
        e.sync()
        t0 = time.perf_counter()
        out = e.admm_run(rho, None, phase1=True, num_iters=iters)
        e.sync()
        dt = time.perf_counter() - t0
        i1, i2 = float(out['iters1'].sum()), float(out['iters2'].sum())
        print('%-32s %.4f s wall, %.3e restart-iterations -> %.3e /s; %.1f + %.1f iterations per restart; one n x n product per phase-2 iteration: %.1f TFLOP/s' % (
            e.last_admm_kernel()[0], dt, i1 + i2, (i1 + i2) / dt, i1 / R, i2 / R, i2 * 2.0 * n * n / dt / 1e12), flush=True)
