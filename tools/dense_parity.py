"""North-star parity statement on the dense-constraint path: the GPU population against the oracle, restart by restart.
Prints, per family: how many restarts leave the oracle's trajectory, the best (objective, max violation) of both
populations, and the distribution of the final objective.  usage: python tools/dense_parity.py [R] [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import oracle as orc
from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm

R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
fams = [('dense n=100 m=30', problems.dense_indefinite(100, 30, seed=11)[0]),
        ('beamforming n=100 (50 antennas, 12+4)', problems.beamforming(50, 12, 4, seed=3)[0]),
        ('dense n=128 m=40', problems.dense_indefinite(128, 40, seed=12)[0])]
# (n = 256, m = 130 costs the faithful oracle 20 s per restart: test_gpu_scale.py samples restarts there)
seed, first = 13, 5
for name, funcs in fams:
    n = funcs[0][0].shape[0]
    e = Engine(QCQPForm.from_arrays(funcs))
    prob = orc.Problem(funcs)
    X0 = 1.5 * np.random.RandomState(3).randn(n, R)
    e.upload(X0)
    out = e.cd_run(phase1=True, num_iters=iters, seed=seed, first_index=first)
    X = e.download()
    f0, mv = e.eval()
    t = time.time()
    Xo = np.zeros_like(X)
    fo = np.zeros(R)
    mo = np.zeros(R)
    for r in range(R):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(first + r)
        x, s1, s2 = prob.improve_cd(X0[:, r], num_iters=iters, rng=rng)
        Xo[:, r] = x
        fo[r] = prob.eval(0, x)
        mo[r] = prob.max_violation(x)
    dt = time.time() - t
    d = np.max(np.abs(X - Xo), axis=0) / (1 + np.max(np.abs(Xo), axis=0))
    same = d < 1e-6
    def best(f, m):
        feas = m < 1e-2      # qcqp.py:252-254 ordering, as better_key
        idx = np.where(feas)[0]
        if len(idx):
            i = idx[np.argmin(f[idx])]
        else:
            i = int(np.argmin(m))
        return i, f[i], m[i]
    ig, fg, mg = best(f0, mv)
    io, fob, mob = best(fo, mo)
    print('%s, R = %d, %d sweeps/phase (oracle %.1f s)' % (name, R, iters, dt))
    print('   restarts on the oracle trajectory (1e-6): %d of %d (%.1f %% diverge); among the diverged: median |dx| %.2e' % (
        same.sum(), R, 100.0 * (1 - same.mean()), np.median(d[~same]) if (~same).any() else 0.0))
    print('   best GPU    restart %4d  f0 %.10f  maxviol %.3e   (same trajectory: %s)' % (ig, fg, mg, bool(same[ig])))
    print('   best oracle restart %4d  f0 %.10f  maxviol %.3e   (same trajectory: %s)' % (io, fob, mob, bool(same[io])))
    print('   relative difference of the best objective %.3e' % (abs(fg - fob) / (1 + abs(fob))))
    print('   feasible (maxviol < 1e-2): GPU %d, oracle %d;  objective quartiles GPU %s  oracle %s' % (
        (mv < 1e-2).sum(), (mo < 1e-2).sum(), np.round(np.percentile(f0, [25, 50, 75]), 4), np.round(np.percentile(fo, [25, 50, 75]), 4)))
    print('   reported vs fresh evaluation: f0 %.2e  maxviol %.2e' % (
        np.max(np.abs(out['f0'] - f0) / (1 + np.abs(f0))), np.max(np.abs(out['maxviol'] - mv))))
