"""BASELINE.json configs[4] flow end to end at a reduced size: suggest(SDR) with the engine's own SDP solver
(general family), Gaussian samples from the relaxation, improve(COORD_DESCENT).   usage: cfg5_sdr.py [n=256] [m=64] [R=512]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import QCQP, SDR, RANDOM, COORD_DESCENT, problems, sdr
from qcqp_amd.api import Problem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
R = int(sys.argv[3]) if len(sys.argv) > 3 else 512
funcs, _, _ = problems.dense_indefinite(n, m, seed=7)
q = QCQP(Problem.from_minimize_form(funcs))
t0 = time.time(); f, v = q.suggest(SDR, num_samples=R, seed=1); t1 = time.time()
info = q.sdr_info
print('cfg5 reduced (n=%d, m=%d): SDP relaxation solved in %.1f s (%d function/gradient evaluations, rank %d, %d outer iterations); '
      'SDR bound %.6g, dual value %.6g' % (n, m, t1 - t0, info['evals'], info['rank'], len(info['hist']), q.sdr_bound, info['dual_value']))
if n <= 1024:
    lmin, S = sdr.dual_certificate_general(q.qcqp_form, info['y'], info['yN'])
    print('   dual certificate: lambda_min(C + sum y_k M_k + y_N E) = %.2e (|S|max %.2e)' % (lmin, np.abs(S).max()))
print('   %d Gaussian samples from the relaxation: best (f, maxviol) = (%.6g, %.3g); feasible samples %d'
      % (R, f, v, (q.population_v < 1e-2).sum()))
t0 = time.time(); f2, v2 = q.improve(COORD_DESCENT, num_iters=30, seed=2); t1 = time.time()
st = q.last_stats
print('   improve(COORD_DESCENT, num_iters=30): %.2f s; best (f, maxviol) = (%.6g, %.3g); feasible restarts %d of %d; '
      'gap to the SDR bound %.3g %%' % (t1 - t0, f2, v2, (st['maxviol'] < 1e-2).sum(), R, 100.0 * (f2 - q.sdr_bound) / abs(q.sdr_bound)))
q2 = QCQP(Problem.from_minimize_form(funcs))
q2.suggest(RANDOM, num_samples=R, seed=1)
f3, v3 = q2.improve(COORD_DESCENT, num_iters=30, seed=2)
print('   same budget from RANDOM starts: best (f, maxviol) = (%.6g, %.3g)' % (f3, v3))
