"""Where the wall time of the full-size SDP relaxation goes on the host side: cProfile of a short solve.
usage: python tools/sdr_hostprofile.py [n=4096] [m=1024] [inner=30]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcqp_amd import problems, sdr
from qcqp_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
inner = int(sys.argv[3]) if len(sys.argv) > 3 else 30
form = problems.dense_indefinite_generated(n, m, seed=7)
e = Engine(form)
e.sync()
use_cprofile = '--cprofile' in sys.argv
pr = cProfile.Profile()
t0 = time.time()
if use_cprofile:
    pr.enable()
X, bound, info = sdr.solve_sdr_general(e, form, outer=1, inner=inner, verbose=True)
if use_cprofile:
    pr.disable()
dt = time.time() - t0
print('%.2f s, %d evaluations, %.1f ms each' % (dt, info['evals'], 1e3 * dt / info['evals']))
print('per evaluation (ms): ' + ', '.join('%s %.1f' % (k, 1e3 * v / info['evals']) for k, v in info['timing'].items()))
if use_cprofile:
    pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
