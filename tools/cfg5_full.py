"""BASELINE.json configs[4] at FULL size on one MI355X: dense random indefinite QCQP n=4096, m=1024
(137.6 GB of fp64 matrices, generated on the device), this GPU's share of the restarts.
usage: cfg5_full.py [R=512] [sweeps=2] [n=4096] [m=1024]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems
from qcqp_amd.engine import Engine

R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
m = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
form = problems.dense_indefinite_generated(n, m, seed=7)
t0 = time.time(); e = Engine(form); e.sync(); t1 = time.time()
gb = (m + 1) * n * n * 8 / 1e9
print('cfg5 full: n=%d m=%d: %.1f GB of matrices generated + packed on the device in %.1f s' % (n, m, gb, t1 - t0))
e.upload(0.1 * np.random.RandomState(0).randn(n, R))
ta = time.time(); f0, mv = e.eval(); tb = time.time()
fl = 2.0 * (m + 1) * n * n
print('eval of %d candidates: %.3f s wall, kernels %.1f ms -> %.1f TFLOP/s (2 (m+1) n^2 flops per candidate); '
      'HBM floor for one pass over the matrices %.1f ms' % (R, tb - ta, e.kernel_ms(0), fl * R / e.kernel_ms(0) / 1e9, gb / 8e3 * 1e3))
print('   start: f0 median %.4g, max violation median %.3g, feasible %d' % (np.median(f0), np.median(mv), (mv < 1e-2).sum()))
ta = time.time(); out = e.cd_run(phase1=True, num_iters=iters, seed=1); tb = time.time()
s1, s2 = out['sweeps1'].sum(), out['sweeps2'].sum()
print('cd_run(num_iters=%d): %.2f s wall; phase 1: %d restart-sweeps in %.1f ms; phase 2: %d restart-sweeps in %.1f ms'
      % (iters, tb - ta, s1, e.kernel_ms(1), s2, e.kernel_ms(2)))
for nm, sw, ms in (('phase 1', s1, e.kernel_ms(1)), ('phase 2', s2, e.kernel_ms(2))):
    if sw:
        print('   %s: %.1f restart-sweeps/s = %.2f TFLOP/s algorithmic (%.1f GFLOP per restart-sweep)' % (nm, sw / ms * 1e3, fl * sw / ms / 1e9, fl / 1e9))
print('   end: f0 median %.4g (start %.4g), feasible %d of %d, accepted moves per restart %.0f'
      % (np.median(out['f0']), np.median(f0), (out['maxviol'] < 1e-2).sum(), R, out['accepted2'].mean()))
