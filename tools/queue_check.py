"""cd_phase2_qs_kernel (restart-level slots + device-side queue) against cd_phase2_q_kernel (tile-bound): same restarts,
per-restart agreement and kernel time.  usage: queue_check.py [n=1024] [R=4096,16384]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
Rs = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [4096, 16384]
funcs, _, _ = problems.boolean_least_squares(n, n // 4, seed=1)
e = Engine(QCQPForm.from_arrays(funcs))
for R in Rs:
    res = {}
    for mode in (0, 1, 1):
        e.cd_queue(mode)
        e.randn(R, seed=2024)
        out = e.cd_run(seed=2024)
        ms = e.kernel_ms(2)
        X = e.download()
        sw = out['visits2'].sum() / float(n)
        print('n=%d R=%d mode %d: %s %.3f ms, %.0f restart-sweeps, %.2f TFLOP/s = %.3f of 78.6; visits max %.1f mean %.1f sweeps'
              % (n, R, mode, e.last_cd_kernel(), ms, sw, sw * 2.0 * n * n / ms / 1e9, sw * 2.0 * n * n / ms / 1e9 / 78.6,
                 out['visits2'].max() / float(n), out['visits2'].mean() / float(n)))
        if mode in res:
            X1, o1 = res[mode]
            print('   repeat: bitwise identical X %s, f0 %s' % (np.array_equal(X, X1), np.array_equal(out['f0'], o1['f0'])))
        res[mode] = (X, out)
    (X0, o0), (X1, o1) = res[0], res[1]
    d = np.max(np.abs(X0 - X1), axis=0) / (1 + np.max(np.abs(X0), axis=0))
    same = all(np.array_equal(o0[k], o1[k]) for k in ('visits2', 'accepted2', 'sweeps2', 'ran_phase2', 'status2'))
    print('   queue vs tile-bound: max|dx| %.2e (restarts above 1e-9: %d), counters identical %s, f0 rel %.2e, maxviol abs %.2e'
          % (d.max(), int((d > 1e-9).sum()), same, np.max(np.abs(o0['f0'] - o1['f0']) / (1 + np.abs(o0['f0']))), np.max(np.abs(o0['maxviol'] - o1['maxviol']))))
