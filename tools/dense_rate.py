"""Dense-constraint coordinate descent (BASELINE.json configs[4] family, matrices generated on the device): restart-sweeps/s and
algorithmic TFLOP/s against the number of restarts.  usage: python tools/dense_rate.py [n] [m] [R ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qcqp_amd import problems
from qcqp_amd.engine import Engine

ARGS = [a for a in sys.argv[1:] if not a.startswith('--')]
n = int(ARGS[0]) if len(ARGS) > 0 else 1024
m = int(ARGS[1]) if len(ARGS) > 1 else 256
Rs = [int(a) for a in ARGS[2:]] or [512, 4096]
form = problems.dense_indefinite_generated(n, m, seed=7)
e = Engine(form)
e.dense_chain_mode(1 if '--one-wave' in sys.argv else 0)
for R in Rs:
    e.randn(R, seed=5)
    e.cd_run(phase1=True, num_iters=1, seed=5)
    e.randn(R, seed=6)
    e.sync()
    t0 = time.perf_counter()
    out = e.cd_run(phase1=True, num_iters=2, seed=6)
    e.sync()
    dt = time.perf_counter() - t0
    sw = float(out['sweeps1'].sum()) + float(out['visits2'].sum()) / n
    fl = sw * 2.0 * n * n * (m + 1)
    print('n %d m %d R %5d: %.3f s, %.0f restart-sweeps/s, %.2f TFLOP/s algorithmic (%.3f of 78.6), %.1f ms per sweep of the population'
          % (n, m, R, dt, sw / dt, fl / 1e12 / dt, fl / 1e12 / dt / 78.6, 1e3 * dt / max(sw / R, 1e-9)))
