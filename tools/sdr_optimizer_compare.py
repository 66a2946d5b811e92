"""The general SDP-relaxation solver with SciPy's L-BFGS-B against the engine's own L-BFGS on the configs[4] family.
usage: python tools/sdr_optimizer_compare.py [n=1024] [m=256] [optimizers=scipy,own]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems, sdr
from qcqp_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = int(sys.argv[2]) if len(sys.argv) > 2 else 256
opts = (sys.argv[3] if len(sys.argv) > 3 else 'scipy,own').split(',')       # own:40:1.5[:sigma0] = own L-BFGS, inner0 40, growth 1.5
form = problems.dense_indefinite_generated(n, m, seed=7)
e = Engine(form)
e.sync()
for opt in opts:
    t0 = time.time()
    parts = opt.split(':')
    kw = dict(inner0=int(parts[1]), growth=float(parts[2])) if len(parts) > 2 else {}
    if len(parts) > 3:
        kw['sigma0'] = float(parts[3])
    if len(parts) > 4:
        kw['ftol_final'] = float(parts[4])
    X, bound, info = sdr.solve_sdr_general(e, form, outer=40, inner=300, optimizer=parts[0], **kw)
    dt = time.time() - t0
    lmin, S = sdr.dual_certificate_device(e, info['y'], info['yN'])
    print('%-12s: %.1f s, %d evaluations (%.1f ms each), %d outer iterations, value %.8g, dual value %.8g, infeasibility %.2e, lambda_min %.2e'
          % (opt, dt, info['evals'], 1e3 * dt / info['evals'], len(info['hist']), bound, info['dual_value'], info['infeas'], lmin), flush=True)
    print('        per evaluation (ms): ' + ', '.join('%s %.1f' % (k, 1e3 * v / info['evals']) for k, v in info['timing'].items()), flush=True)
