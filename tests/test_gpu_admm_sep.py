"""improve_admm (qcqp.py:195-285) for SEPARABLE constraints -- p x_i^2 + q x_i + r ~ 0: Boolean least squares
(/root/reference/examples/boolean_least_squares.py:34-36), MAXCUT (maxcut.py:25-28), boxes -- through bases of unit vectors
(qcqpmi_admm_unit_bases, round 5): the eigenvectors utilities.py:160-162 takes from LAPACK for P_k = p e_i e_i^T ARE unit
vectors, so the bases are written down (QCQPForm.unit_bases) and the two consensus products of an iteration become a gather and
a scatter.  Checked against the oracle's improve_admm (which decomposes every constraint matrix with LAPACK like the reference),
against the GEMM path on the same bases, and through the public API.  Tolerance: 1e-6, the north star's for ADMM."""
import numpy as np
import pytest

from conftest import oracle_map

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng_mod():
    from qcqp_amd import engine
    assert engine.device_count() >= 1, 'no HIP device visible'
    return engine


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b)) / (1.0 + np.abs(np.asarray(b))))


def family(name, n):
    from qcqp_amd import problems
    if name == 'bls':
        return problems.boolean_least_squares(n, max(n // 4, 4), seed=3)[0]
    if name == 'box':
        return problems.box_least_squares(n, max(n // 4, 4), seed=3)[0]
    if name == 'maxcut':
        return problems.maxcut(n, 0.5, seed=3, weighted=True)[0]
    if name == 'lin2':
        # TWO constraints per coordinate, both linear (P_k = 0: the basis vector carries the q direction only, eigenvalue 0):
        # x_i <= u_i and -x_i <= -l_i around a least-squares objective -- the scatter sums two rows per coordinate
        rs = np.random.RandomState(4)
        A = rs.randn(max(n // 2, 4), n)
        b = rs.randn(max(n // 2, 4)) * np.sqrt(n)
        fs = [(A.T.dot(A) + 0.1 * np.eye(n), -2.0 * A.T.dot(b), float(b.dot(b)), None)]
        lo, hi = -0.5 - rs.rand(n), 0.5 + rs.rand(n)
        for i in range(n):
            e = np.zeros(n); e[i] = 1.0
            fs.append((np.zeros((n, n)), e, -hi[i], '<='))
            fs.append((np.zeros((n, n)), -e, lo[i], '<='))
        return fs
    raise ValueError(name)


def rho_for(funcs):
    """improve_admm's automatic rho (qcqp.py:270-277)."""
    P0 = np.asarray(funcs[0][0].todense()) if hasattr(funcs[0][0], 'todense') else np.asarray(funcs[0][0])
    lmin = np.linalg.eigvalsh((P0 + P0.T) / 2.0)[0]
    m = len(funcs) - 1
    return 50.0 * (2.0 * (1.0 - lmin) / m if lmin < 0 else 1.0 / m)


def run_engine(eng_mod, form, bases, rho, X0, iters, unit, f0_by_product=False, every_step=False, three_launches=False):
    lam, Bv, qhat = bases
    e = eng_mod.Engine(form)
    if f0_by_product or every_step or three_launches:
        # debug bit 4: f0(z) of every phase-2 iterate through the product with P0; bit 8: every step of the bisection evaluates phi;
        # bit 2: gather, projection and scatter of an iteration as three launches instead of admm_unit_step_kernel
        e.L.qcqpmi_debug_profile(e.h, ((4 if f0_by_product else 0) | (8 if every_step else 0) | (2 if three_launches else 0)) << 4, None)
    assert e.separable
    e.admm_set_basis(lam, Bv, qhat)
    e.admm_unit_bases(unit)
    res, its = e.admm_zsolver_device(rho)
    assert res < 1e-9
    e.upload(X0)
    out = e.admm_run(rho, None, phase1=True, num_iters=iters)
    X = e.download()
    f0, mv = e.eval()
    assert rel(out['f0'], f0) < 1e-9 and np.max(np.abs(out['maxviol'] - mv)) < 1e-9
    assert e.last_admm_kernel()[0] == ('admm_multi_launch<unit bases>' if unit else 'admm_multi_launch')
    e.close()
    return X, out


@pytest.mark.parametrize('name,n,R', [('bls', 48, 40), ('bls', 100, 24), ('box', 64, 24), ('maxcut', 40, 24), ('lin2', 32, 24)])
def test_admm_unit_bases_vs_oracle_and_gemm_path(eng_mod, orc, name, n, R):
    """Two runs against the oracle's improve_admm (LAPACK eigenpairs of every constraint, like the reference):
      * 10 + 10 iterations: every sampled restart within 1e-9 -- the iteration is the reference's;
      * 30 + 30 iterations: median within 1e-6, every one within 1e-3.  onecons_qcqp bisects the multiplier to 1e-6 (utilities.py:149, 187-194) and the
        projection onto x_i^2 = 1 switches sign where z_i + u_i crosses zero: a comparison that rounding decides the other way
        moves an iterate by the solver's own tolerance and later iterates amplify it (measured with the FULL eigenbasis as
        well: 1e-11 after 10 iterations, 2e-6 on one restart in eight after 30, 1e-5 on several after 60) -- the reference fixes those restarts no better.
    And the gather / scatter path against the GEMM path on the same bases: identical iteration counts, 1e-9."""
    from qcqp_amd.form import QCQPForm
    funcs = family(name, n)
    form = QCQPForm.from_arrays(funcs)
    assert form.m == (2 * n if name == 'lin2' else n)
    ub = form.unit_bases()
    assert ub is not None
    lam = ub[0]
    # what LAPACK returns for these matrices: the eigenvalues are EXACTLY {0, ..., p} (so the bracket of the multiplier,
    # utilities.py:176-180, is the one the nonzero eigenvalue gives) and the eigenvectors are unit vectors
    for k in (0, n // 2, form.m - 1):
        Pk = np.asarray(form.fs[k].P.todense()) if hasattr(form.fs[k].P, 'todense') else np.asarray(form.fs[k].P)
        w, Q = np.linalg.eigh((Pk + Pk.T) / 2.0)
        assert sorted(w.tolist()) == sorted([0.0] * (n - 1) + [lam[k, 0]])
        assert np.all(np.sort(np.abs(Q), axis=0)[-1] == 1.0) and np.count_nonzero(Q) == n
    rho = rho_for(funcs)
    X0 = np.random.RandomState(11).randn(n, R)
    prob = orc.Problem(funcs)
    sample = (0, 1, 2, R // 2, R - 2, R - 1)
    for iters in (10, 30):
        Xu, ou = run_engine(eng_mod, form, ub, rho, X0, iters, True)
        Xg, og = run_engine(eng_mod, form, ub, rho, X0, iters, False)
        d = np.max(np.abs(Xu - Xg), axis=0) / (1 + np.max(np.abs(Xg), axis=0))
        assert d.max() < 1e-9
        assert np.array_equal(ou['iters1'], og['iters1']) and np.array_equal(ou['iters2'], og['iters2'])
        dev = np.array([rel(Xu[:, r], xa) for r, xa in
                        zip(sample, oracle_map(lambda r: prob.improve_admm(X0[:, r], num_iters=iters, rho=rho), sample))])
        print('\nADMM unit bases, %s n=%d R=%d, %d + %d iterations (rho %.3g): vs the GEMM path max %.1e; vs the oracle on %d restarts '
              'median %.1e max %.1e' % (name, n, R, iters, iters, rho, d.max(), len(sample), np.median(dev), dev.max()))
        if iters == 10:
            assert dev.max() < 1e-9
        else:
            assert np.median(dev) < 1e-6 and dev.max() < 1e-3


def test_admm_objective_of_the_iterates_from_the_solve(eng_mod):
    """Phase 2 needs f0(z) of every iterate for `better` (qcqp.py:249).  z solves 2 (P0 + rho m I) z = rhs, so
    P0 z = rhs / 2 - rho m z: the engine takes f0 from the solve's own right-hand side instead of a second n x n product per
    iteration.  Same iterates, same bookkeeping as with the product (debug bit 4), on a family with a dense P0."""
    from qcqp_amd.form import QCQPForm
    for name, n, R, iters in (('bls', 100, 48, 40), ('maxcut', 60, 32, 40)):
        funcs = family(name, n)
        form = QCQPForm.from_arrays(funcs)
        ub = form.unit_bases()
        rho = rho_for(funcs)
        X0 = np.random.RandomState(3).randn(n, R)
        Xa, oa = run_engine(eng_mod, form, ub, rho, X0, iters, True)
        Xb, ob = run_engine(eng_mod, form, ub, rho, X0, iters, True, f0_by_product=True)
        d = np.max(np.abs(Xa - Xb), axis=0) / (1 + np.max(np.abs(Xb), axis=0))
        print('\nf0 of the iterates from the solve vs through P0, %s n=%d: max|dx| %.1e, f0 %.1e' % (name, n, d.max(), rel(oa['f0'], ob['f0'])))
        assert d.max() < 1e-9 and rel(oa['f0'], ob['f0']) < 1e-9
        assert np.array_equal(oa['iters1'], ob['iters1']) and np.array_equal(oa['iters2'], ob['iters2'])


@pytest.mark.parametrize('name,n', [('bls', 100), ('box', 64), ('maxcut', 40), ('lin2', 32)])
def test_admm_unit_step_kernel_is_the_three_launches(eng_mod, name, n):
    """admm_unit_step_kernel (round 5): ZQ = s Z[i] (gather), the projection of every (constraint, restart) pair
    (onecons_qcqp, utilities.py:149-196) and S[i] = sum s D (scatter) of one ADMM iteration in one launch, a thread per
    (coordinate, restart) -- against the three launches it replaces: the same points, objectives, violations and
    iteration counts, bit for bit (65 restarts: a partial tile; `lin2`: two constraints on every coordinate)."""
    from qcqp_amd.form import QCQPForm
    funcs = family(name, n)
    form = QCQPForm.from_arrays(funcs)
    ub = form.unit_bases()
    rho = rho_for(funcs)
    X0 = np.random.RandomState(23).randn(n, 65) * 2.0
    Xa, oa = run_engine(eng_mod, form, ub, rho, X0, 40, True)
    Xb, ob = run_engine(eng_mod, form, ub, rho, X0, 40, True, three_launches=True)
    assert np.array_equal(Xa, Xb)
    assert np.array_equal(oa['iters1'], ob['iters1']) and np.array_equal(oa['iters2'], ob['iters2'])
    assert np.array_equal(oa['f0'], ob['f0']) and np.array_equal(oa['maxviol'], ob['maxviol'])


@pytest.mark.parametrize('name,n', [('bls', 100), ('box', 64), ('maxcut', 40), ('lin2', 32)])
def test_admm_one_row_bisection_shortcut_is_the_bisection(eng_mod, name, n):
    """admm_secular_small_kernel<1> decides the comparisons of onecons_qcqp's loops (utilities.py:176-194) by the side of the
    closed-form root the trial multiplier lies on and evaluates the secular function only near it.  Same multipliers, same
    points: the run with the shortcut equals the run that evaluates every step, bit for bit."""
    from qcqp_amd.form import QCQPForm
    funcs = family(name, n)
    form = QCQPForm.from_arrays(funcs)
    ub = form.unit_bases()
    rho = rho_for(funcs)
    X0 = np.random.RandomState(17).randn(n, 64) * 2.0
    Xa, oa = run_engine(eng_mod, form, ub, rho, X0, 40, True)
    Xb, ob = run_engine(eng_mod, form, ub, rho, X0, 40, True, every_step=True)
    assert np.array_equal(Xa, Xb)
    assert np.array_equal(oa['iters1'], ob['iters1']) and np.array_equal(oa['iters2'], ob['iters2'])
    assert np.array_equal(oa['f0'], ob['f0'])


def test_improve_admm_on_boolean_least_squares_through_the_api(eng_mod, orc):
    """QCQP.improve(ADMM) on the reference's own example family takes the unit-bases setup by itself; one start, the
    reference's defaults but for the iteration count; against the oracle."""
    from test_gpu_api import handler
    from qcqp_amd import ADMM
    funcs = family('bls', 36)
    q = handler(funcs)
    x0 = np.random.RandomState(5).randn(36)
    q.prob.variables()[0].value = x0.reshape(-1, 1)
    f, v = q.improve(ADMM, num_iters=80)
    assert q.last_stats['setup'] == 'unit bases (separable constraints)'
    xa = orc.Problem(funcs).improve_admm(x0, num_iters=80)
    xg = np.ravel(q.prob.variables()[0].value)
    assert rel(xg, xa) < 1e-6


def test_improve_admm_on_maxcut_through_the_api(eng_mod, orc):
    """MAXCUT (maxcut.py:25-28: maximise, an indefinite objective in minimise form) through QCQP.improve(ADMM): the automatic rho
    of improve_admm for lambda_min < 0 (qcqp.py:270-277: Lanczos on the device here, LAPACK in the reference), unit bases,
    20 iterations per phase; the oracle runs the minimise form with the rho the handler chose."""
    from test_gpu_api import handler
    from qcqp_amd import ADMM, problems
    funcs, _, _ = problems.maxcut(30, 0.5, seed=2, weighted=True)
    q = handler(funcs, maximize=True)
    x0 = np.random.RandomState(9).randn(30)
    q.prob.variables()[0].value = x0.reshape(-1, 1)
    q.improve(ADMM, num_iters=20)
    assert q.last_stats['setup'] == 'unit bases (separable constraints)'
    rho = q.last_stats['rho']
    assert abs(rho - rho_for(funcs)) <= 1e-6 * rho          # Lanczos' lambda_min against LAPACK's
    xa = orc.Problem(funcs).improve_admm(x0, num_iters=20, rho=rho)
    assert rel(np.ravel(q.prob.variables()[0].value), xa) < 1e-6


def test_admm_unit_bases_full_size_boolean_least_squares(eng_mod):
    """n = 1024, m = 1024 (BASELINE.json configs[1]'s problem), 512 restarts, 30 + 30 iterations: what the reference's setup
    makes of this size is 1024 LAPACK decompositions of 1024 x 1024 matrices and 8.6 GB of eigenvectors; here the bases are a
    table.  Size-independent checks: the gather / scatter path equals the GEMM path on the same bases, the reported (f0, max
    violation) are those of the returned points, and ADMM never returns a point that `better` ranks below its start."""
    from qcqp_amd import problems
    from qcqp_amd.form import QCQPForm
    n, R, iters = 1024, 512, 30
    funcs, _, _ = problems.boolean_least_squares(n, n // 4, seed=1)
    form = QCQPForm.from_arrays(funcs)
    lam, Bv, qhat = form.unit_bases()
    rho = rho_for(funcs)
    X0 = np.random.RandomState(2).randn(n, R)
    outs = []
    for unit in (True, False):
        e = eng_mod.Engine(form)
        e.admm_set_basis(lam, Bv, qhat)
        e.admm_unit_bases(unit)
        e.admm_zsolver_device(rho)
        e.upload(X0)
        f00, mv00 = e.eval()
        out = e.admm_run(rho, None, phase1=True, num_iters=iters)
        X = e.download()
        f0, mv = e.eval()
        assert rel(out['f0'], f0) < 1e-9 and np.max(np.abs(out['maxviol'] - mv)) < 1e-9
        b0, b1 = np.floor(mv00 / 1e-4), np.floor(mv / 1e-4)
        assert np.all((b1 < b0) | ((b1 == b0) & (f0 <= f00 * (1 + 1e-12) + 1e-9)))       # utilities.py:138-146
        outs.append(X)
        e.close()
    d = np.max(np.abs(outs[0] - outs[1]), axis=0) / (1 + np.max(np.abs(outs[1]), axis=0))
    print('\nADMM unit bases at n = 1024, m = 1024, R = %d: vs the GEMM path max|dx| median %.2e max %.2e' % (R, np.median(d), d.max()))
    assert d.max() < 1e-9


def test_admm_kernels_random_shapes():
    """tools/fuzz_admm.py with a fixed seed: admm_fused_kernel in both geometries against the multi-launch path (1e-6, equal
    iteration counts) on random beamforming shapes, admm_unit_step_kernel against the launches it replaces (bit for bit) on
    random separable problems -- 12 + 12 cases beyond the fixed shapes of the tests above (improve_admm, qcqp.py:254-285)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, 'tools', 'fuzz_admm.py'), '12', '7'], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert 'mismatches: 0' in p.stdout, p.stdout[-3000:]
    assert p.stdout.count('\nB ') + p.stdout.startswith('B ') == 12 and 'identical' in p.stdout
