"""Stage split of dense_chain_kernel (the wave of restart 0, s_memtime ticks) on the configs[4] family.
usage: python tools/dense_stage_profile.py [n=1024] [m=256] [R=512]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems
from qcqp_amd.engine import Engine

ARGS = [a for a in sys.argv[1:] if not a.startswith('--')]
n = int(ARGS[0]) if len(ARGS) > 0 else 1024
m = int(ARGS[1]) if len(ARGS) > 1 else 256
R = int(ARGS[2]) if len(ARGS) > 2 else 512
form = problems.dense_indefinite_generated(n, m, seed=7)
e = Engine(form)
e.dense_chain_mode(1 if '--one-wave' in sys.argv else 0)
e.randn(R, seed=5)
e.cd_run(phase1=True, num_iters=1, seed=5)
e.L.qcqpmi_debug_profile(e.h, 1, None)
e.randn(R, seed=6)
out = e.cd_run(phase1=True, num_iters=2, seed=6)
pr = np.zeros(64, dtype=np.int64)
e.L.qcqpmi_debug_dense_profile(e.h, pr.ctypes.data_as(C.POINTER(C.c_int64)))
names = ['set-up', 'coefficients', 'bounds+reduce', 'gaps', 'segment sweep', 'minimise/draw', 'commit', 'write-back']
for ph in (0, 1, 2, 3):
    p = pr[16 * ph:16 * ph + 16]
    if not p[:10].any():
        continue
    tot = float(p[:8].sum())
    nc = max(1, int(p[8]))
    print('%s phase %d: %d coordinates, %d feasible-set evaluations, %.0f ticks per coordinate (%.2f us at 2.1 GHz); ' % ('serial thread' if ph < 2 else 'thread 0', ph % 2 + 1, nc, p[9], tot / nc, tot / nc / 2100.0)
          + ', '.join('%s %.0f (%.0f%%)' % (nm, p[k] / nc, 100.0 * p[k] / max(tot, 1.0)) for k, nm in enumerate(names)))
