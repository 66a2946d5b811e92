"""BASELINE.json configs[4] at FULL size, the whole flow on one MI355X: the engine's own SDP relaxation of the
137.6 GB problem (matrices generated on the device), dual certificate from the device-assembled dual matrix,
Gaussian samples, coordinate descent.   usage: cfg5_sdr_full.py [n=4096] [m=1024] [R=512] [outer=25] [inner=300]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems, sdr
from qcqp_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
R = int(sys.argv[3]) if len(sys.argv) > 3 else 512
outer = int(sys.argv[4]) if len(sys.argv) > 4 else 25
inner = int(sys.argv[5]) if len(sys.argv) > 5 else 300
form = problems.dense_indefinite_generated(n, m, seed=7)
t0 = time.time(); e = Engine(form); e.sync(); t1 = time.time()
print('n=%d m=%d: %.1f GB of matrices on the device in %.1f s' % (n, m, (m + 1) * n * n * 8 / 1e9, t1 - t0), flush=True)
t0 = time.time(); X, bound, info = sdr.solve_sdr_general(e, form, outer=outer, inner=inner, verbose=True); t1 = time.time()
print('SDP relaxation: %.1f s, %d evaluations (%.0f ms each), rank %d, %d outer iterations; value %.6g, dual value %.6g'
      % (t1 - t0, info['evals'], 1e3 * (t1 - t0) / info['evals'], info['rank'], len(info['hist']), bound, info['dual_value']), flush=True)
print('   per evaluation (ms): ' + ', '.join('%s %.1f' % (k, 1e3 * v / info['evals']) for k, v in info['timing'].items()), flush=True)
t0 = time.time(); lmin, S = sdr.dual_certificate_device(e, info['y'], info['yN']); t1 = time.time()
print('dual certificate (dual matrix assembled on the device, eigvalsh on the host %.1f s): lambda_min %.3e (|S|max %.2e)' % (t1 - t0, lmin, np.abs(S).max()), flush=True)
# samples from N(mu, Sigma) of the relaxation, then coordinate descent
mu = X[:n, n].copy()
Sigma = X[:n, :n] - np.outer(mu, mu) + 1e-8 * np.eye(n)
w, U = np.linalg.eigh(Sigma)
F = U * np.sqrt(np.maximum(w, 0.0))
e.sdr_sample(mu, F, R, seed=1)
f0, mv = e.eval()
print('%d samples: best feasible objective %s, feasible %d' % (R, ('%.6g' % f0[mv < 1e-2].min()) if (mv < 1e-2).any() else 'none', (mv < 1e-2).sum()), flush=True)
t0 = time.time(); out = e.cd_run(phase1=True, num_iters=6, seed=2); t1 = time.time()
ok = out['maxviol'] < 1e-2
print('improve(COORD_DESCENT, num_iters=6): %.1f s; feasible %d of %d; best objective %.6g; gap to the certified bound %.2f %%'
      % (t1 - t0, ok.sum(), R, out['f0'][ok].min(), 100.0 * (out['f0'][ok].min() - bound) / abs(bound)))
