"""Wall time of the three device calls one function / gradient evaluation of the relaxation solver makes, at full size.
usage: python tools/sdr_evaltime.py [n=4096] [m=1024] [rank=47]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems
from qcqp_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
r = int(sys.argv[3]) if len(sys.argv) > 3 else 47
form = problems.dense_indefinite_generated(n, m, seed=7)
e = Engine(form)
e.sync()
V = 0.1 * np.random.RandomState(0).randn(n, r)
w = np.random.RandomState(1).rand(m + 1)
pause = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0      # seconds of host idle between evaluations
for rep in range(6):
    time.sleep(pause)
    t0 = time.perf_counter(); e.upload(V); t1 = time.perf_counter()
    quad, lin = e.eval_parts(); t2 = time.perf_counter()
    Y = e.weighted_product(w); t3 = time.perf_counter()
    print('upload %.1f ms, eval_parts %.1f ms, weighted_product %.1f ms' % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
