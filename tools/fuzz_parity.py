"""Randomised parity sweep: coordinate descent on the GPU (pipelined / general kernels) against the oracle
with the shared keyed stream, over many small problems of the separable families.  Not a test (runtime),
a shake-out tool: prints every mismatch.   usage: fuzz_parity.py [cases=120] [seed=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems
from qcqp_amd.engine import Engine, EngineError
from qcqp_amd.form import QCQPForm
from oracle import oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(cases):
    fam = rs.choice(['bls', 'bls_scaled', 'maxcut', 'box'])
    n = int(rs.choice([5, 12, 16, 17, 31, 32, 33, 48, 64, 80, 96]))     # 48..96: the quad-chain kernel (NB >= 3)
    if fam == 'bls':
        funcs, _, _ = problems.boolean_least_squares(n, max(2, n // 2 + int(rs.randint(0, n))), seed=int(rs.randint(1 << 30)))
    elif fam == 'bls_scaled':     # x_i^2 == d_i with a different d per coordinate (one class per coordinate)
        funcs, _, _ = problems.boolean_least_squares(n, n, seed=int(rs.randint(1 << 30)))
        funcs = [funcs[0]] + [(P * (1.0 + i % 3), q, r * (1.0 + i % 3) * (0.5 + 0.1 * (i % 7)), rel) for i, (P, q, r, rel) in enumerate(funcs[1:])]
    elif fam == 'maxcut':
        funcs, _, _ = problems.maxcut(n, 0.5, seed=int(rs.randint(1 << 30)), weighted=True)
    else:                          # box constraints lo <= x_i <= hi as one concave quadratic each, PSD objective
        G = rs.randn(n, n)
        funcs = [(G.T.dot(G) / n + 0.1 * np.eye(n), rs.randn(n), 0.0, None)]
        for i in range(n):
            P = np.zeros((n, n)); P[i, i] = 1.0
            funcs.append((P, np.zeros(n), -(0.5 + rs.rand()), '<='))
    R = int(rs.choice([1, 3, 16, 21]))
    seed, first = int(rs.randint(1 << 20)), int(rs.randint(100))
    X0 = (0.3 + 1.5 * rs.rand()) * rs.randn(n, R)
    e = Engine(QCQPForm.from_arrays(funcs))
    prob = orc.Problem(funcs)
    e.upload(X0)
    try:
        out = e.cd_run(phase1=True, num_iters=40, seed=seed, first_index=first)
    except EngineError as err:
        print('case %d %s n=%d: engine error %s' % (case, fam, n, err)); bad += 1; continue
    X = e.download()
    for r in range(R):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(first + r)
        x, s1, s2 = prob.improve_cd(X0[:, r], num_iters=40, rng=rng)
        d = np.max(np.abs(X[:, r] - x) / (1 + np.abs(x)))
        if d > 1e-9 or out['visits2'][r] != s2[1] or out['accepted2'][r] != s2[2] or out['sweeps1'][r] != s1[0]:
            print('case %d %s n=%d R=%d restart %d: max rel diff %.3e visits %d/%d accepted %d/%d sweeps1 %d/%d'
                  % (case, fam, n, R, r, d, out['visits2'][r], s2[1], out['accepted2'][r], s2[2], out['sweeps1'][r], s1[0]))
            bad += 1
    e.close()
print('%d cases, %d mismatching restarts' % (cases, bad))
