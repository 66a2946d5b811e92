import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from qcqp_amd import problems, engine
from qcqp_amd.form import QCQPForm
from oracle import oracle
n, R, iters = 48, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 60
funcs = problems.boolean_least_squares(n, 12, seed=3)[0]
form = QCQPForm.from_arrays(funcs)
lam, Bv, qhat = form.unit_bases()
P0 = np.asarray(funcs[0][0]); m = n
lmin = np.linalg.eigvalsh(P0)[0]
rho = 50.0 * (1.0 / m)
X0 = np.random.RandomState(11).randn(n, R)
def rel(a, b): return np.max(np.abs(a - b) / (1 + np.abs(b)))
prob = oracle.Problem(funcs)
for ph1 in (True, False):
    xa = [prob.improve_admm(X0[:, r], num_iters=iters, rho=rho, phase1=ph1) for r in range(R)]
    for mode in ('unit+ns', 'unit+inv', 'eig+inv'):
        e = engine.Engine(form)
        if mode.startswith('unit'):
            e.admm_set_basis(lam, Bv, qhat)
        else:
            lm = np.zeros((m, n)); Q = np.zeros((m, n, n))
            for k, f in enumerate(form.fs):
                lm[k], Q[k] = np.linalg.eigh(np.asarray(f.P.todense()) if hasattr(f.P, 'todense') else np.asarray(f.P))
            e.admm_set_eig(lm, Q)
        Minv = None
        if mode.endswith('ns'):
            print('  ns', e.admm_zsolver_device(rho))
        else:
            Minv = np.linalg.inv(2.0 * (P0 + rho * m * np.eye(n)))
        e.upload(X0)
        out = e.admm_run(rho, Minv, phase1=ph1, num_iters=iters)
        X = e.download()
        print('phase1 %s %s: vs oracle %s iters1 %s iters2 %s' % (ph1, mode, ['%.1e' % rel(X[:, r], xa[r]) for r in range(R)], out['iters1'][:4], out['iters2'][:4]))
        e.close()
