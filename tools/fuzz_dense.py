"""Randomised sweep of the dense-constraint path (forced also for n <= 64) against the oracle: small dense indefinite
and beamforming problems, random shapes.  Dense family: points must agree up to O(tol) threshold flips; both
families: reported values = oracle evaluation of the returned point, same feasibility class.   usage: fuzz_dense.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems
from qcqp_amd.engine import Engine, EngineError
from qcqp_amd.form import QCQPForm
from oracle import oracle as orc

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
bad = flips = tot = 0
stat = {'dense': [0, 0, []], 'beam': [0, 0, []]}
for case in range(cases):
    fam = rs.choice(['dense', 'beam'])
    if fam == 'dense':
        n, m = int(rs.choice([6, 16, 20, 33, 48])), int(rs.choice([1, 3, 7, 12]))
        funcs, _, _ = problems.dense_indefinite(n, m, seed=int(rs.randint(1 << 30)))
        X0s = 0.3 + rs.rand()
    else:
        na, mh, l = int(rs.choice([3, 6, 10])), int(rs.choice([1, 2, 3])), int(rs.choice([1, 2]))
        funcs, _, _ = problems.beamforming(na, mh, l, seed=int(rs.randint(1 << 30)))
        n, m, X0s = 2 * na, mh + l, 2.0
    R = int(rs.choice([1, 5, 16, 19]))
    iters = int(rs.choice([1, 4, 12]))
    seed, first = int(rs.randint(1 << 20)), int(rs.randint(50))
    X0 = X0s * rs.randn(n, R)
    if only >= 0 and case != only:
        continue
    e = Engine(QCQPForm.from_arrays(funcs))
    if not (len(sys.argv) > 4 and sys.argv[4] == 'exact'):   # 'exact': n <= 64 stays on the reference-arithmetic kernel
        e.L.qcqpmi_debug_profile(e.h, 32 << 4, None)
    prob = orc.Problem(funcs)
    e.upload(X0)
    try:
        out = e.cd_run(phase1=True, num_iters=iters, seed=seed, first_index=first)
    except EngineError as err:
        # the reference raises on these too?  (unbounded set with zero objective in phase 1)
        try:
            rng = orc.Rng(orc.RNG_KEYED, seed); rng.set_restart(first)
            prob.improve_cd(X0[:, 0], num_iters=iters, rng=rng)
            print('case %d %s: engine error only: %s' % (case, fam, str(err)[:80])); bad += 1
        except RuntimeError:
            pass
        e.close(); continue
    X = e.download()
    for r in range(R):
        tot += 1
        rng = orc.Rng(orc.RNG_KEYED, seed); rng.set_restart(first + r)
        try:
            x, s1, s2 = prob.improve_cd(X0[:, r], num_iters=iters, rng=rng)
        except RuntimeError:
            continue
        d = np.max(np.abs(X[:, r] - x))
        ok_val = abs(prob.eval(0, X[:, r]) - out['f0'][r]) <= 1e-9 * (1 + abs(out['f0'][r])) and abs(prob.max_violation(X[:, r]) - out['maxviol'][r]) < 1e-9
        same_class = (out['maxviol'][r] < 1e-2) == (prob.max_violation(x) < 1e-2)
        if not ok_val or (fam == 'dense' and (d > 1e-3 or not same_class)):
            print('case %d %s n=%d m=%d R=%d it=%d restart %d: |dx| %.2e ok_val %s same_class %s' % (case, fam, n, m, R, iters, r, d, ok_val, same_class)); bad += 1
            if only >= 0:
                idx = np.argsort(-np.abs(X[:, r] - x))[:6]
                print('   coords', idx, 'gpu', X[idx, r], 'oracle', x[idx], 'x0', X0[idx, r])
                print('   gpu f0 %.10g viol %.3e | oracle f0 %.10g viol %.3e; sweeps1 gpu %d oracle %d' % (out['f0'][r], out['maxviol'][r], prob.eval(0, x), prob.max_violation(x), out['sweeps1'][r], s1[0]))
        elif d > 1e-9:
            flips += 1
        stat[fam][0] += 1
        if d > 1e-9:
            stat[fam][1] += 1; stat[fam][2].append(d)
    e.close()
print('%d cases, %d restarts, %d bad, %d with O(tol)/chaotic differences' % (cases, tot, bad, flips))
for fam in stat:
    dd = np.array(stat[fam][2]) if stat[fam][2] else np.zeros(1)
    print('   %s: %d restarts, %d differ > 1e-9 (median %.1e, 90%% %.1e, max %.1e)' % (fam, stat[fam][0], stat[fam][1], np.median(dd), np.percentile(dd, 90), dd.max()))
