import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from conftest import funcs_from_npz, load_golden, RELSTR
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm
from oracle import oracle as orc
z = load_golden('g5_onecons')
n = z['P'].shape[1]
bad = 0
for i in range(z['P'].shape[0]):
    funcs = [(np.eye(n), np.zeros(n), 0., None), (z['P'][i], z['q'][i], float(z['r'][i]), RELSTR[int(z['relop'][i])])]
    e = Engine(QCQPForm.from_arrays(funcs))
    e.admm_set_eig(z['lmb'][i][None], z['Q'][i][None])
    e.upload(np.stack([z['z'][i]] * 3, axis=1))
    x = e.admm_onecons(1)
    d = np.max(np.abs(x - z['x'][i][:, None]))
    if d > 1e-9:
        bad += 1
        print('g5 case', i, 'early', bool(z['early'][i]), 'diff', d, x[:3, 0], z['x'][i][:3])
print('g5 onecons mismatches:', bad, 'of', z['P'].shape[0])
g = load_golden('g8_admm_beam10')
funcs = funcs_from_npz(g)
n = int(g['n']); m = len(funcs) - 1
prob = orc.Problem(funcs)
rho = float(g['rho'])
P0 = np.asarray(funcs[0][0])
Minv = np.linalg.inv(2. * (P0 + rho * m * np.eye(n)))
for iters in (1, 2, 5, int(g['iters'])):
    e = Engine(QCQPForm.from_arrays(funcs))
    e.admm_set_eig(g['lmb'], g['Q'])
    e.upload(np.stack([g['x0']] * 2, axis=1))
    out = e.admm_run(rho, Minv, phase1=True, num_iters=iters)
    X = e.download()
    xa = prob.improve_admm(g['x0'], num_iters=iters, rho=rho)
    print('iters', iters, 'gpu vs oracle', np.max(np.abs(X[:, 0] - xa)), 'it1', out['iters1'], 'it2', out['iters2'], 'f0', out['f0'][0], prob.eval(0, xa))
