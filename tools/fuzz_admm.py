"""Randomised shake-out of the ADMM kernels of round 5 (not a test: runtime; prints every mismatch).
  A. admm_fused_kernel in both geometries (eight-wave workgroups; four-wave workgroups two per CU, qcqpmi_admm_fused(ctx, 2)) against
     the multi-launch path on random beamforming problems (reduced bases of rank 2, diagonal P0): points within 1e-6 (median 1e-9),
     iteration counts equal on > 97 % of the restarts -- the assertions of tests/test_gpu_scale.py on shapes the tests do not visit.
  B. admm_unit_step_kernel (z-update + gather + projection + scatter in one launch) against the launches it replaces (debug bit 2)
     on random separable problems (Boolean least squares, box, weighted MAXCUT): bit-identical points, objectives and counts.
usage: fuzz_admm.py [cases=30] [seed=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import lowrank, problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


# ---------------------------------------------------------------- A: fused kernel, both geometries, vs multi-launch
for case in range(cases):
    nant = int(rs.choice([24, 32, 40, 48, 64, 96, 128, 200]))
    mh, ml = int(rs.choice([2, 3, 6, 8])), int(rs.choice([3, 5, 10, 20]))
    R = int(rs.choice([1, 7, 16, 33, 96, 200, 512]))
    iters = int(rs.choice([5, 20, 60]))
    funcs, _, _ = problems.beamforming(nant, mh, ml, seed=int(rs.randint(1, 1000)))
    form = QCQPForm.from_arrays(funcs)
    e = Engine(form)
    rb = lowrank.reduced_bases(e, form)
    if rb is None:
        e.close()
        continue
    lam, Bv, qhat, info = rb
    e.admm_set_basis(lam, Bv, qhat)
    X0 = rs.randn(form.n, R)
    res = {}
    for mode in (1, 2, 0):
        e.admm_fused(mode)
        e.upload(X0)
        if mode == 1:
            res['p1'] = bool(rs.rand() < 0.8)
        out = e.admm_run(1.0, None, phase1=res['p1'], num_iters=iters)
        res[mode] = (e.download(), out, e.last_admm_kernel())
    Xm, om, _ = res[0]
    line = 'A %3d: n=%4d m=%3d R=%4d iters=%3d phase1=%d' % (case, form.n, form.m, R, iters, res['p1'])
    for mode in (1, 2):
        X, o, (name, cw) = res[mode]
        d = np.max(np.abs(X - Xm), axis=0) / (1 + np.max(np.abs(Xm), axis=0))
        same = np.mean((o['iters1'] == om['iters1']) & (o['iters2'] == om['iters2']))
        ok = name == 'admm_fused_kernel' and np.median(d) < 1e-9 and d.max() < 1e-6 and same > 0.97 or (R < 40 and d.max() < 1e-6 and same >= 1.0 - 1.5 / R)
        line += ' | mode %d C=%2d: max|dx| %.1e same counts %.3f%s' % (mode, cw, d.max(), same, '' if ok else '  <-- MISMATCH')
        bad += 0 if ok else 1
    print(line, flush=True)
    e.close()

# ---------------------------------------------------------------- B: unit step vs the launches it replaces
for case in range(cases):
    fam = str(rs.choice(['bls', 'box', 'maxcut']))
    n = int(rs.choice([16, 31, 48, 64, 100, 160, 256]))
    R = int(rs.choice([1, 15, 16, 17, 65, 300]))
    iters = int(rs.choice([5, 25, 50]))
    seed = int(rs.randint(1, 1000))
    if fam == 'bls':
        funcs = problems.boolean_least_squares(n, max(n // 4, 4), seed=seed)[0]
    elif fam == 'box':
        funcs = problems.box_least_squares(n, max(n // 4, 4), seed=seed)[0]
    else:
        funcs = problems.maxcut(n, 0.5, seed=seed, weighted=True)[0]
    form = QCQPForm.from_arrays(funcs)
    ub = form.unit_bases()
    P0 = np.asarray(funcs[0][0].todense()) if hasattr(funcs[0][0], 'todense') else np.asarray(funcs[0][0])
    lmin = np.linalg.eigvalsh((P0 + P0.T) / 2.0)[0]
    m = len(funcs) - 1
    rho = max(50.0 / m, -2.0 * lmin / m + 1e-6) if lmin < 0 else 50.0 / m
    X0 = rs.randn(n, R) * 2.0
    outs = []
    for three in (False, True):
        e = Engine(form)
        if three:
            e.L.qcqpmi_debug_profile(e.h, 2 << 4, None)
        e.admm_set_basis(*ub)
        e.admm_zsolver_device(rho)
        e.upload(X0)
        out = e.admm_run(rho, None, phase1=True, num_iters=iters)
        outs.append((e.download(), out))
        e.close()
    (Xa, oa), (Xb, ob) = outs
    ok = (np.array_equal(Xa, Xb) and np.array_equal(oa['iters1'], ob['iters1']) and np.array_equal(oa['iters2'], ob['iters2'])
          and np.array_equal(oa['f0'], ob['f0']) and np.array_equal(oa['maxviol'], ob['maxviol']))
    print('B %3d: %-6s n=%4d R=%4d iters=%3d: %s' % (case, fam, n, R, iters, 'identical' if ok else 'MISMATCH max|dx| %.2e' % np.max(np.abs(Xa - Xb))), flush=True)
    bad += 0 if ok else 1
print('mismatches: %d' % bad)
