"""Second-generation phase-2 kernel (cd_phase2_q.h) against the first generation (cd_phase2_rs.h) and the oracle.
usage: python tools/q_check.py [n] [R]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
funcs, _, _ = problems.boolean_least_squares(n, n // 4, seed=1)
e = Engine(QCQPForm.from_arrays(funcs))
res = {}
for name, mode in (('rs (gen 1)', 64 << 4), ('q cs=0', (128 | (0 << 8)) << 4), ('q cs=2', (128 | (2 << 8)) << 4),
                   ('q cs=4', (128 | (4 << 8)) << 4), ('q cs=6', (128 | (6 << 8)) << 4)):
    e.L.qcqpmi_debug_profile(e.h, mode, None)
    for rep in range(3):
        e.randn(R, seed=5)
        out = e.cd_run(seed=5)
    X = e.download()
    f0, mv = e.eval()
    res[name] = (X, out, e.kernel_ms(2))
    sw = out['visits2'].sum() / n
    print('%-12s phase2 %.3f ms  sweeps %.0f  -> %.2f TFLOP/s  tracked-vs-fresh f0 %.2e  maxviol diff %.2e' % (
        name, res[name][2], sw, sw * 2.0 * n * n / res[name][2] / 1e9, np.max(np.abs(out['f0'] - f0) / (1 + np.abs(f0))),
        np.max(np.abs(out['maxviol'] - mv))))
X0, o0, _ = res['rs (gen 1)']
for name in res:
    X, o, _ = res[name]
    d = np.max(np.abs(X - X0), axis=0)
    print('%-12s vs gen 1: restarts identical to 1e-9: %d of %d, max diff %.2e, visits equal %d, best f0 %.10f / %.10f' % (
        name, int((d < 1e-9).sum()), R, d.max(), int((o['visits2'] == o0['visits2']).sum()), o['f0'].min(), o0['f0'].min()))
if n <= 256:
    from oracle import oracle as orc
    prob = orc.Problem(funcs)
    X0s = orc.keyed_normal_matrix(5, n, 8, first_index=0)
    X, o, _ = res['q cs=4']
    for r in range(8):
        rng = orc.Rng(orc.RNG_KEYED, 5)
        rng.set_restart(r)
        x, s1, s2 = prob.improve_cd(X0s[:, r], rng=rng)
        print('oracle restart', r, 'max |dx| %.2e' % np.max(np.abs(x - X[:, r])), 'visits', s2[1], o['visits2'][r], 'acc', s2[2], o['accepted2'][r])
