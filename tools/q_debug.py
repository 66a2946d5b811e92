import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import funcs_from_npz, load_golden
from oracle import oracle as orc
z = load_golden('g6_cd_bls32')
funcs = funcs_from_npz(z)
prob = orc.Problem(funcs)
for iters in (1, 2, 3):
    for mode, nm in ((64 << 4, 'gen1'), (0, 'q'), (1 << 4, 'q lockstep')):
        e = Engine(QCQPForm.from_arrays(funcs))
        e.L.qcqpmi_debug_profile(e.h, mode, None)
        e.upload(z['X0'])
        out = e.cd_run(phase1=False, num_iters=iters)
        X = e.download()
        ds = []
        for r in range(X.shape[1]):
            x, st = prob.cd_phase2(z['X0'][:, r], num_iters=iters, rng=orc.Rng(orc.RNG_KEYED, 0))
            d = np.abs(X[:, r] - x)
            ds.append((d.max(), int(np.argmax(d > 1e-9)) if (d > 1e-9).any() else -1, int(out['visits2'][r]), int(st[1])))
        print(nm, 'iters', iters, ds)
