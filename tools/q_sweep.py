"""Regression sweep of the quad-chain phase-2 kernel against the first-generation kernel over every supported n:
n = 48, 64, ..., 1024 (multiples of 16), a ragged number of restarts each.  usage: python tools/q_sweep.py [step=16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm

step = int(sys.argv[1]) if len(sys.argv) > 1 else 16
bad = 0
count = 0
for n in range(48, 1025, step):
    R = 17 + (n * 7) % 40
    funcs, _, _ = problems.boolean_least_squares(n, max(4, n // 4), seed=n)
    e = Engine(QCQPForm.from_arrays(funcs))
    res = {}
    for name, mode in (('rs', 64 << 4), ('q', 0)):
        e.L.qcqpmi_debug_profile(e.h, mode, None)
        e.randn(R, seed=n + 1)
        out = e.cd_run(seed=n + 1)
        res[name] = (e.download(), out, e.last_cd_kernel())
    Xr, outr, kr = res['rs']
    Xq, outq, kq = res['q']
    d = np.max(np.abs(Xr - Xq))
    ok = d < 1e-9 and np.array_equal(outr['visits2'], outq['visits2']) and np.array_equal(outr['accepted2'], outq['accepted2']) \
        and np.max(np.abs(outr['f0'] - outq['f0']) / (1 + np.abs(outr['f0']))) < 1e-9
    count += 1
    if not ok or kq != 'cd_phase2_q_kernel' or kr != 'cd_phase2_rs_kernel':
        bad += 1
        print('n %4d R %3d: kernels %s / %s, max |dx| %.2e, visits equal %s' % (n, R, kr, kq, d, np.array_equal(outr['visits2'], outq['visits2'])))
print('%d sizes, %d mismatches' % (count, bad))
