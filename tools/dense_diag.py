"""Diagnostic: dense path vs oracle, one phase-1 sweep, first coordinate where a restart leaves the oracle's trajectory.
usage: python tools/dense_diag.py [n] [m] [R] [family: dense | beam]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm
from oracle import oracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = int(sys.argv[2]) if len(sys.argv) > 2 else 130
R = int(sys.argv[3]) if len(sys.argv) > 3 else 16
seed, first = 13, 5
fam = sys.argv[4] if len(sys.argv) > 4 else 'dense'
if fam == 'beam':
    funcs, _, _ = problems.beamforming(n // 2, m, max(m // 3, 1), seed=3)
else:
    funcs, _, _ = problems.dense_indefinite(n, m, seed=11)
e = Engine(QCQPForm.from_arrays(funcs))
prob = orc.Problem(funcs)
X0 = 1.5 * np.random.RandomState(3).randn(n, R)
for p1 in (1, 2):
    e.upload(X0)
    e.L.qcqpmi_debug_profile(e.h, 0, None)
    out = e.cd_run(phase1=True, num_iters=p1, viol_tol=-1.0, seed=seed, first_index=first)   # viol_tol < 0: the gate never opens -> phase 1 only
    X = e.download()
    nbad = 0
    for r in range(R):
        rng = orc.Rng(orc.RNG_KEYED, seed); rng.set_restart(first + r)
        x, st = prob.cd_phase1(X0[:, r], num_iters=p1, viol_tol=-1.0, rng=rng)
        d = np.abs(X[:, r] - x)
        bad = np.nonzero(d > 1e-9)[0]
        nbad += len(bad) > 0
        if len(bad):
            i = bad[0]
            print('phase-1 sweeps %d restart %2d: max diff %.2e, first coordinate off: %d (diff %.2e), then %s' % (
                p1, r, d.max(), i, d[i], ' '.join('%.1e' % v for v in d[i:i + 6])))
    print('phase-1 sweeps %d: %d of %d restarts leave the oracle trajectory (> 1e-9)' % (p1, nbad, R))
