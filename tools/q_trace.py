"""Event trace of one workgroup (tile 0) of the phase-2 kernel cd_phase2_q_kernel: when each wave waits, multiplies,
publishes.  usage: python tools/q_trace.py [n] [R] [cs] [first] [count]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import numpy as np

from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
cs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
first = int(sys.argv[4]) if len(sys.argv) > 4 else 20
count = int(sys.argv[5]) if len(sys.argv) > 5 else 8
funcs, _, _ = problems.boolean_least_squares(n, n // 4, seed=1)
e = Engine(QCQPForm.from_arrays(funcs))
mode = (128 | (cs << 8)) << 4
e.L.qcqpmi_debug_profile(e.h, mode | 1, None)
e.randn(R, seed=2)
e.cd_run()
tr = np.zeros(2048, dtype=np.int64)
rc = e.L.qcqpmi_debug_trace(e.h, tr.ctypes.data_as(C.POINTER(C.c_int64)), 2048)
assert rc == 0
tr = tr.reshape(8, 32, 8)
t0 = tr[0, first, 0]
for k in range(first, min(first + count, 32)):
    c = tr[0, k] - t0
    print('block %2d chain: top %7d  tiles ready %7d (+%5d)  commit %7d (+%5d)  fix-up done %7d (+%5d)   staging published %7d' % (
        k, c[0], c[1], c[1] - c[0], c[2], c[2] - c[1], c[3], c[3] - c[2], tr[4, k, 0] - t0))
    for w in ((1, 2, 3) if k % 2 == 0 else (5, 6, 7)):
        p = tr[w, k] - t0
        print('    wave %d product %2d: top %7d  commit seen %7d  refreshed %7d  mfmas issued %7d (+%5d)  slot free %7d (+%5d)  stored %7d (+%5d)  published %7d (+%5d)' % (
            w, k, p[0], p[1], p[2], p[3], p[3] - p[2], p[4], p[4] - p[3], p[5], p[5] - p[4], p[6], p[6] - p[5]))
