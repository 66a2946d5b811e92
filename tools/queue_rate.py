"""Kernel time of phase 2 with the tile-bound kernel (mode 0), the slot-queue kernel forced (1) and the default dispatch (2), every
mode warmed up first, several seeds.  usage: queue_rate.py [n=1024] [R=16384,65536] [reps=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
Rs = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [16384, 65536]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
funcs, _, _ = problems.boolean_least_squares(n, n // 4, seed=1)
e = Engine(QCQPForm.from_arrays(funcs))
for R in Rs:
    for mode in (0, 1, 2, 0, 1):
        e.cd_queue(mode)
        e.randn(R, seed=90)
        e.cd_run(phase1=True, seed=90)              # warm-up of this mode
        ms = sw = 0.0
        per = []
        for k in range(reps):
            e.randn(R, seed=91 + k)
            out = e.cd_run(phase1=True, seed=91 + k)
            t = e.kernel_ms(2)
            ms += t; per.append(t)
            sw += out['visits2'].sum() / float(n)
        print('n=%d R=%d mode %d: %-20s %.3f ms per launch (%s), %.2f TFLOP/s = %.3f of 78.6'
              % (n, R, mode, e.last_cd_kernel(), ms / reps, ' '.join('%.2f' % t for t in per), sw * 2.0 * n * n / ms / 1e9, sw * 2.0 * n * n / ms / 1e9 / 78.6), flush=True)
