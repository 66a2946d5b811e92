"""GPU tests of the drop-in facade (qcqp_amd.QCQP) against the API-level golden vectors."""
import numpy as np
import pytest

from conftest import funcs_from_npz, load_golden

pytestmark = pytest.mark.gpu


def handler(funcs, maximize=False):
    from qcqp_amd import QCQP, Problem
    return QCQP(Problem.from_minimize_form(funcs, maximize=maximize))


def test_constants_and_star_import():
    ns = {}
    exec('from qcqp_amd import *', ns)
    for name in ['QCQP', 'RANDOM', 'SPECTRAL', 'SDR', 'COORD_DESCENT', 'ADMM', 'DCCP', 'IPOPT']:
        assert name in ns


def test_suggest_random_and_phase2_match_reference_api_flow():
    """G10: np.random.seed(42); suggest(RANDOM) gives the reference's point and (f, v); phase 2 run
    from the reference's coordinate-descent result is a fixed point with the reference's (f, v)."""
    from qcqp_amd import RANDOM, COORD_DESCENT
    z = load_golden('g10_api_bls10')
    q = handler(funcs_from_npz(z))
    np.random.seed(int(z['seed']))
    f, v = q.suggest(RANDOM)
    x = np.ravel(q.prob.variables()[0].value, order='F')
    assert np.array_equal(x, z['x_rand'])
    assert abs(f - z['fv'][0, 0]) <= 1e-12 * (1 + abs(f)) and abs(v - z['fv'][0, 1]) <= 1e-13
    q.prob.variables()[0].value = z['x_cd'].reshape(-1, 1)
    f, v = q.improve(COORD_DESCENT, phase1=False)
    assert abs(f - z['fv'][1, 0]) <= 1e-9 * (1 + abs(f))
    assert abs(v - z['fv'][1, 1]) <= 1e-12
    assert np.max(np.abs(np.ravel(q.prob.variables()[0].value) - z['x_cd'])) < 1e-9


@pytest.mark.parametrize('name', ['bls10', 'maxcut12'])
def test_suggest_sdr_matches_reference_draws(name):
    """G9: injected lifted solution, np.random.seed(7): same draws, same (f, v) incl. the sign flip
    of maximisation problems."""
    from qcqp_amd import SDR
    z = load_golden('g9_sdr_' + name)
    q = handler(funcs_from_npz(z), maximize=bool(z['maximize']))
    np.random.seed(int(z['seed']))
    for t in range(z['xs'].shape[1]):
        f, v = q.suggest(SDR, X=z['X']) if t == 0 else q.suggest(SDR)
        x = np.ravel(q.prob.variables()[0].value, order='F')
        assert np.max(np.abs(x - z['xs'][:, t])) < 1e-12
        assert abs(f - z['fv'][t, 0]) <= 1e-11 * (1 + abs(f))
        assert abs(v - z['fv'][t, 1]) <= 1e-11 * (1 + abs(v))
    assert np.max(np.abs(q.Sigma - z['Sigma'])) < 1e-15


def test_population_flow_best_of_restarts(orc):
    """suggest(RANDOM, num_samples=R) + improve(COORD_DESCENT): best (f, v) equals the oracle's best
    over the same keyed starts and draws."""
    from qcqp_amd import RANDOM, COORD_DESCENT, problems, dist
    funcs, _, _ = problems.boolean_least_squares(40, 24, seed=8)
    q = handler(funcs)
    R, seed = 48, 5
    q.suggest(RANDOM, num_samples=R, seed=seed)
    X0 = q.population()
    f, v = q.improve(COORD_DESCENT, seed=seed)
    prob = orc.Problem(funcs)
    fs, vs = [], []
    for r in range(R):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(r)
        x, _, _ = prob.improve_cd(X0[:, r], rng=rng)
        fs.append(prob.eval(0, x))
        vs.append(prob.max_violation(x))
    key = dist.select_best_host(fs, vs, 1e-4)
    assert q.best_index == key[2]
    assert abs(f - key[1]) <= 1e-9 * (1 + abs(f))
    assert np.max(np.abs(q.population_f - np.array(fs)) / (1 + np.abs(np.array(fs)))) < 1e-9


def test_errors_mirror_reference():
    from qcqp_amd import ADMM, SDR, problems
    funcs, _, _ = problems.boolean_least_squares(6, 8, seed=1)
    q = handler(funcs)
    with pytest.raises(Exception) as ei:
        q.suggest('nope')
    assert 'Unknown suggest method' in str(ei.value.args[0])
    with pytest.raises(Exception) as ei:
        q.improve('nope')
    assert 'Unknown improve method(s)' in str(ei.value.args[0])
    q.suggest(SDR)              # Boolean family: the engine's own SDP solver applies
    assert q.sdr_sol is not None and q.sdr_bound is not None
    with pytest.raises(Exception) as ei:
        q.improve('dccp')
    assert 'DCCP package is not installed.' in str(ei.value)


def test_suggest_sdr_separable_families():
    """solve_sdr (qcqp.py:72-97) for separable constraints other than x_i^2 = d_i: boxes / discs on single
    coordinates go through the Burer-Monteiro solver with elementwise constraint operators (P0 V on the device).
    Convex case with a known answer: min sum x_i^2 + x_i  s.t. x_i^2 <= 2  ->  x = -1/2, value -n/4, relaxation
    tight; mixed case: an indefinite objective with boxes and an annulus, bound certified (dual slack PSD) and below
    every feasible point that local search finds."""
    from qcqp_amd import SDR, COORD_DESCENT
    n = 12
    funcs = [(np.eye(n), np.ones(n), 0.0, None)]
    for i in range(n):
        P = np.zeros((n, n)); P[i, i] = 1.0
        funcs.append((P, np.zeros(n), -2.0, '<='))
    q = handler(funcs)
    np.random.seed(0)
    f, v = q.suggest(SDR)
    assert q.sdr_info['converged'] and q.sdr_info.get('family') == 'separable'
    assert abs(q.sdr_bound - (-0.25 * n)) < 1e-5
    assert np.max(np.abs(q.mu - (-0.5))) < 1e-4
    # indefinite objective, boxes -1 <= x_i <= 2 written as (x - 1/2)^2 <= 9/4, one annulus 1 <= x_0^2 <= 4
    rs = np.random.RandomState(3)
    G = rs.randn(n, n); G = (G + G.T) / 2
    funcs = [(G, rs.randn(n), 0.0, None)]
    for i in range(n):
        P = np.zeros((n, n)); P[i, i] = 1.0
        qq = np.zeros(n); qq[i] = -1.0
        funcs.append((P, qq, -2.0, '<='))          # x^2 - x - 2 <= 0  <=>  -1 <= x <= 2
    P = np.zeros((n, n)); P[0, 0] = -1.0
    funcs.append((P, np.zeros(n), 1.0, '<='))      # 1 - x_0^2 <= 0
    q = handler(funcs)
    f, v = q.suggest(SDR, num_samples=64, seed=1)
    assert q.sdr_info['lambda_min'] > -1e-5 * (1 + abs(q.sdr_bound))
    xs = np.array(q._assigned, copy=True)
    pf, pv = np.array(q.population_f, copy=True), np.array(q.population_v, copy=True)
    # the same 64 samples drawn + evaluated in ONE call without a resident population (qcqpmi_sdr_sample_eval): same values, same
    # winner, the winner re-drawn from its index is the resident point the improve() below starts from
    g, w = q.suggest(SDR, num_samples=64, seed=1, keep_population=False)
    assert (g, w) == (f, v) and np.array_equal(q._assigned, xs)
    assert np.array_equal(q.population_f, pf) and np.array_equal(q.population_v, pv)
    assert q.engine.pop_size == 1
    f2, v2 = q.improve(COORD_DESCENT, seed=2)
    assert v2 < 1e-2 and q.sdr_bound <= f2 + 1e-6 * (1 + abs(f2))


def test_improve_method_list_with_admm_matches_reference():
    """G10, third call: improve([COORD_DESCENT, ADMM], phase1=False, num_iters=50) started from the
    reference's coordinate-descent point reproduces the reference's (f, v) and point (same kwargs go to
    both methods, auto-rho, phase-1 skipped: everything deterministic)."""
    from qcqp_amd import COORD_DESCENT, ADMM
    z = load_golden('g10_api_bls10')
    q = handler(funcs_from_npz(z))
    q.prob.variables()[0].value = z['x_cd'].reshape(-1, 1)
    f, v = q.improve([COORD_DESCENT, ADMM], phase1=False, num_iters=50)
    assert abs(f - z['fv'][2, 0]) <= 1e-6 * (1 + abs(f))
    assert abs(v - z['fv'][2, 1]) <= 1e-6
    assert np.max(np.abs(np.ravel(q.prob.variables()[0].value) - z['x_chain'])) < 1e-6


def test_admm_rho_too_small_raises_like_reference():
    from qcqp_amd import ADMM, RANDOM, problems
    funcs, _, _ = problems.maxcut(8, 0.5, seed=1)      # indefinite objective
    q = handler(funcs, maximize=True)
    q.suggest(RANDOM)
    with pytest.raises(Exception) as ei:
        q.improve(ADMM, rho=1e-6)
    assert 'rho parameter is too small' in str(ei.value)


# ------------------------------------------------------------- own SDP relaxation (solve_sdr)
def _brute_force_pm1(P0, q0, r0, n):
    best = np.inf
    for mask in range(1 << n):
        x = np.array([1.0 if (mask >> i) & 1 else -1.0 for i in range(n)])
        best = min(best, x.dot(P0.dot(x)) + q0.dot(x) + r0)
    return best


def test_sdr_solver_unit_diagonal_family_optimality():
    """solve_sdr replacement (mixing method on the device): no reference result exists to compare with
    (third-party solver), so the solution is validated by optimality conditions -- dual certificate
    S = C + diag(y) PSD, zero duality gap -- and by brute force on small instances:
    SDP bound <= optimum (BLS, minimise);  optimum <= SDP bound <= optimum / 0.878 (MAXCUT)."""
    from qcqp_amd import problems, sdr
    from qcqp_amd.engine import Engine
    from qcqp_amd.form import QCQPForm
    # Boolean least squares, the README data (np.random.seed(1), n=10, 15 rows): optimum 35.55097
    funcs, _, _ = problems.boolean_least_squares(10, 15, seed=1)
    form = QCQPForm.from_arrays(funcs)
    e = Engine(form)
    X, bound, info = sdr.solve_sdr(e, form)
    y, lmin, lower = sdr.dual_certificate(info['C'], info['V'])
    assert lmin > -1e-6 * (1 + np.abs(info['C']).max())
    assert abs(bound - (-y.sum())) <= 1e-6 * (1 + abs(bound))            # zero duality gap
    assert np.allclose(np.diag(X), 1.0, atol=1e-12)
    assert np.linalg.eigvalsh(X)[0] > -1e-9
    P0, q0, r0 = np.asarray(funcs[0][0]), np.asarray(funcs[0][1]), funcs[0][2]
    opt = _brute_force_pm1(P0, q0, r0, 10)
    assert lower <= opt + 1e-9 and bound <= opt + 1e-6
    assert bound > 0.5 * opt                                                   # and not a trivial bound
    assert np.all(np.diff(info['hist'][:-1]) <= 1e-9)                         # monotone decrease
    # MAXCUT n=12 (maximise): cut <= SDP bound <= cut / 0.878
    funcs, maxi, ex = problems.maxcut(12, 0.5, seed=3)
    form = QCQPForm.from_arrays(funcs)
    e = Engine(form)
    X, bound, info = sdr.solve_sdr(e, form)
    y, lmin, lower = sdr.dual_certificate(info['C'], info['V'])
    assert lmin > -1e-6
    P0, q0, r0 = np.asarray(funcs[0][0]), np.asarray(funcs[0][1]), funcs[0][2]
    best_cut = -_brute_force_pm1(P0, q0, r0, 12)
    sdp_cut = -bound
    assert best_cut <= sdp_cut + 1e-6 and sdp_cut <= best_cut / 0.878 + 1e-6


def test_suggest_sdr_end_to_end_without_external_solver():
    """BASELINE configs[0]: suggest(SDR) + improve(COORD_DESCENT) on the README problem, the SDP solved by
    the engine itself; with samples from the relaxation the best point reaches the brute-force optimum."""
    from qcqp_amd import QCQP, SDR, COORD_DESCENT, problems
    from qcqp_amd.api import Problem
    funcs, _, _ = problems.boolean_least_squares(10, 15, seed=1)
    prob = Problem.from_minimize_form(funcs)
    q = QCQP(prob)
    f, v = q.suggest(SDR, num_samples=64, seed=11)
    assert q.sdr_bound is not None and q.sdr_bound <= 35.55097 + 1e-4
    f, v = q.improve(COORD_DESCENT, seed=3)
    assert v < 1e-2
    x = np.sign(prob.variables()[0].value).ravel()
    P0, q0, r0 = np.asarray(funcs[0][0]), np.asarray(funcs[0][1]), funcs[0][2]
    fx = x.dot(P0.dot(x)) + q0.dot(x) + r0
    assert abs(fx - 35.55097) < 1e-3, fx
    assert q.sdr_bound <= fx + 1e-6


def test_sdr_general_solver_certified_and_consistent_with_mixing():
    """solve_sdr for ANY QCQP of the dense path (Burer-Monteiro + augmented Lagrangian, constraint values and
    gradients on the device).  (1) dense indefinite family: primal feasible, dual certificate PSD, primal value
    = dual value -y_N; (2) a Boolean problem made non-separable by an inactive coupling constraint has the SAME
    SDP as its separable twin: the general solver must land on the mixing method's optimum."""
    from qcqp_amd import problems, sdr
    from qcqp_amd.engine import Engine
    from qcqp_amd.form import QCQPForm
    funcs, _, _ = problems.dense_indefinite(24, 6, seed=11)
    form = QCQPForm.from_arrays(funcs)
    e = Engine(form)
    e.L.qcqpmi_debug_profile(e.h, 32 << 4, None)       # dense path at n <= 64
    X, bound, info = sdr.solve_sdr_general(e, form)
    lmin, S = sdr.dual_certificate_general(form, info['y'], info['yN'])
    assert lmin > -1e-6 * (1 + np.abs(S).max())
    assert abs(bound - info['dual_value']) <= 1e-5 * (1 + abs(bound))
    assert np.all(info['y'][1:] >= 0.0)                 # inequality multipliers
    n = 24
    for k, f in enumerate(funcs[1:]):                   # primal feasibility of the lifted solution
        P, q, r = np.asarray(f[0]), np.asarray(f[1]), f[2]
        val = np.sum(P * X[:n, :n]) + q.dot(X[:n, n]) + r
        assert val <= 1e-5 * (1 + abs(r)), (k, val)
    assert abs(X[n, n] - 1.0) < 1e-9 and np.linalg.eigvalsh(X)[0] > -1e-9
    # any feasible point of the QCQP is above the bound
    x = np.zeros(n)
    assert funcs[0][2] >= bound - 1e-9 or any(f[2] > 0 for f in funcs[1:])
    # (2) Boolean least squares + inactive ball constraint sum x_i^2 <= 2 n (couples the coordinates)
    n = 10
    fb, _, _ = problems.boolean_least_squares(n, 15, seed=1)
    sep_form = QCQPForm.from_arrays(fb)
    X0, b0, _ = sdr.solve_sdr(Engine(sep_form), sep_form)
    G = np.ones((n, n)) * 1e-3 + np.eye(n)              # dense, PSD, inactive: x'Gx <= 3 n
    fc = list(fb) + [(G, np.zeros(n), -3.0 * n, '<=')]
    form2 = QCQPForm.from_arrays(fc)
    e2 = Engine(form2)
    assert not e2.separable
    e2.L.qcqpmi_debug_profile(e2.h, 32 << 4, None)
    X2, b2, info2 = sdr.solve_sdr_general(e2, form2)
    assert abs(b2 - b0) <= 1e-5 * (1 + abs(b0)), (b2, b0)
    assert abs(info2['y'][-1]) < 1e-6                    # the inactive constraint has a zero multiplier
    assert np.max(np.abs(np.diag(X2)[:n] - 1.0)) < 1e-5


def test_sdr_general_solver_optimizers_agree():
    """The relaxation solver's own L-BFGS loop with inexact inner solves (the default since round 3: 5.5x faster at full
    size, profiles/r03_cfg5_sdr.md) against SciPy's L-BFGS-B with full inner solves on the dense family at n = 128, m = 24:
    the same certified bound from both, the timing breakdown of the solve is reported."""
    from qcqp_amd import problems, sdr
    from qcqp_amd.engine import Engine
    form = problems.dense_indefinite_generated(128, 24, seed=3)
    e = Engine(form)
    res = {}
    for opt, kw in (('own', {}), ('scipy', dict(inner0=None))):
        X, bound, info = sdr.solve_sdr_general(e, form, optimizer=opt, **kw)
        lmin, S = sdr.dual_certificate_device(e, info['y'], info['yN'])
        assert lmin > -1e-6 * (1 + np.abs(S).max()), (opt, lmin)
        assert abs(bound - info['dual_value']) <= 2e-5 * (1 + abs(bound)), (opt, bound, info['dual_value'])
        assert info['infeas'] < 1e-6
        assert set(info['timing']) >= {'upload', 'eval_parts', 'weighted_product', 'total', 'optimizer_and_rest'}
        res[opt] = (bound, info['evals'])
    assert abs(res['own'][0] - res['scipy'][0]) <= 2e-5 * (1 + abs(res['scipy'][0])), res
    print('\nSDP relaxation n=128 m=24: own L-BFGS + inexact inner solves %d evaluations, L-BFGS-B %d; bound %.8g' % (res['own'][1], res['scipy'][1], res['own'][0]))


def test_suggest_spectral_relaxation():
    """suggest(SPECTRAL) (qcqp.py:41-70, 383-388): for the Boolean family the relaxation is the sphere-constrained
    problem  min x'P0x + q0'x + r0  s.t.  ||x||^2 = n  (all equalities summed) -- a trust-region subproblem with a
    rank-one SDP solution: the returned point satisfies its KKT conditions  (P0 + mu I) x = -q0/2,  P0 + mu I PSD,
    and the value is a lower bound of the SDR bound and of the optimum."""
    from qcqp_amd import QCQP, SPECTRAL, SDR, problems
    from qcqp_amd.api import Problem
    funcs, _, _ = problems.boolean_least_squares(10, 15, seed=1)
    P0, q0, r0 = np.asarray(funcs[0][0]), np.asarray(funcs[0][1]), funcs[0][2]
    n = 10
    q = QCQP(Problem.from_minimize_form(funcs))
    f, v = q.suggest(SPECTRAL)
    x = np.asarray(q.spectral_sol)
    fx = x.dot(P0.dot(x)) + q0.dot(x) + r0
    assert abs(abs(x.dot(x)) - n) < 1e-4 * n                       # on the sphere (sign of the eigenvector is free)
    Xs = q.spectral_info['X']
    assert np.linalg.eigvalsh(Xs)[-2] < 1e-4 * np.linalg.eigvalsh(Xs)[-1]   # rank one
    xs = Xs[:n, n]                                                   # the relaxation's point with the right sign
    g = 2.0 * P0.dot(xs) + q0
    mu = -g.dot(xs) / (2.0 * xs.dot(xs))
    assert np.linalg.norm(g + 2.0 * mu * xs) <= 1e-3 * (1 + np.linalg.norm(g))
    assert np.linalg.eigvalsh(P0 + mu * np.eye(n))[0] > -1e-4
    assert abs(q.spectral_bound - (xs.dot(P0.dot(xs)) + q0.dot(xs) + r0)) <= 1e-4 * (1 + abs(q.spectral_bound))
    q.suggest(SDR)
    assert q.spectral_bound <= q.sdr_bound + 1e-5 <= 35.55097 + 1e-3   # spectral <= SDR <= optimum
    assert f == pytest.approx(fx, rel=1e-9) or f == pytest.approx(x.dot(P0.dot(x)) + q0.dot(x) + r0, rel=1e-9)


def test_api_on_device_generated_problem():
    """The drop-in API on a problem that only exists on the device (problems.GeneratedForm, the cfg5 family):
    suggest(SDR) solves the relaxation with the general solver (linear terms come back from the context),
    samples, improve(COORD_DESCENT) -- and the result respects the certified bound."""
    from qcqp_amd import QCQP, SDR, COORD_DESCENT, problems, sdr
    form = problems.dense_indefinite_generated(40, 6, seed=21)
    q = QCQP(form)
    f, v = q.suggest(SDR, num_samples=64, seed=3)
    lmin, S = sdr.dual_certificate_device(q.engine, q.sdr_info['y'], q.sdr_info['yN'])
    assert lmin > -1e-6 * (1 + np.abs(S).max())
    assert abs(q.sdr_bound - q.sdr_info['dual_value']) <= 1e-5 * (1 + abs(q.sdr_bound))
    f2, v2 = q.improve(COORD_DESCENT, num_iters=20, seed=1)
    assert v2 < 1e-2 and f2 >= q.sdr_bound - 1e-2 * (1 + abs(q.sdr_bound))
    assert q.prob.variables()[0].value.shape == (40, 1)


def test_cvxpy_adapter_drives_the_gpu_path():
    """SURVEY 8f-2 on the device: a duck-typed cvxpy >= 1 problem -- the README example as one writes it in cvxpy,
    minimize sum_squares(A x - b) s.t. square(x) == 1 (examples/boolean_least_squares.py:6-15; get_qcqp_form,
    utilities.py:318-347) -- goes through qcqp_amd.cvxpy_adapter into the HIP engine: QCQP(problem) ->
    suggest(RANDOM) -> improve(COORD_DESCENT) -> improve([COORD_DESCENT, ADMM], phase1=False) must return what the
    raw-array path returns BIT FOR BIT (the extracted coefficients are exact for this data up to the rounding of the
    probe evaluations, so the form is compared first), the reference's golden (f, v) of G10 are met, and the cvxpy
    variable holds the final point in its own shape."""
    from duck_cvxpy import Var, Expr, Objective, Equality, Prob
    from qcqp_amd import QCQP, RANDOM, COORD_DESCENT, ADMM
    z = load_golden('g10_api_bls10')
    funcs = funcs_from_npz(z)
    n = funcs[0][0].shape[0]
    P0, q0, r0 = np.asarray(funcs[0][0]), np.asarray(funcs[0][1]), float(funcs[0][2])
    x = Var((n,))
    obj = Expr(lambda: float(x.value.dot(P0).dot(x.value) + q0.dot(x.value) + r0), ())
    cons = [Equality(Expr(lambda: x.value ** 2 - 1.0, (n,)))]
    qa = QCQP(Prob(Objective('minimize', obj), cons, [x]))
    qr = handler(funcs)
    fa, fr = qa.qcqp_form, qr.qcqp_form
    assert (fa.n, fa.m) == (fr.n, fr.m)
    assert np.max(np.abs(np.asarray(fa.f0.P) - np.asarray(fr.f0.P))) < 1e-12 * np.max(np.abs(P0))
    # identical coefficients from here on: the comparison below is about the PATH (adapter -> variables -> engine)
    qa2 = QCQP(Prob(Objective('minimize', obj), cons, [x]))
    qa2.qcqp_form.f0.P[...] = np.asarray(fr.f0.P)
    qa2.qcqp_form.f0.qarray[...] = fr.f0.qarray
    qa2.qcqp_form.f0.r = fr.f0.r
    qa2 = QCQP(qa2.prob)            # engine rebuilt on the patched form
    res = []
    for q in (qa2, qr):
        np.random.seed(int(z['seed']))
        out = [q.suggest(RANDOM)]
        out.append(q.improve(COORD_DESCENT, seed=11))
        out.append(q.improve([COORD_DESCENT, ADMM], phase1=False, seed=12))
        res.append((out, np.ravel(q.prob.variables()[0].value, order='F').copy()))
    (oa, xa), (orr, xr) = res
    assert oa == orr and np.array_equal(xa, xr)
    assert abs(oa[0][0] - z['fv'][0, 0]) <= 1e-12 * (1 + abs(oa[0][0])) and abs(oa[0][1] - z['fv'][0, 1]) <= 1e-13
    assert x.value.shape == (n,) and np.array_equal(x.value, xa)      # the cvxpy variable itself carries the point
    # the adapter's own extraction (no patching) ends on the same point to rounding
    np.random.seed(int(z['seed']))
    qa.suggest(RANDOM)
    f1, v1 = qa.improve(COORD_DESCENT, seed=11)
    assert abs(f1 - oa[1][0]) <= 1e-9 * (1 + abs(f1)) and abs(v1 - oa[1][1]) <= 1e-9


@pytest.mark.parametrize('n', [64, 40])
def test_suggest_batches_streams_improve(n):
    """Population streaming behind the drop-in API: the reference's user loop `for ...: suggest(); improve()` (README.md:51-57)
    for K batches at once -- suggest(RANDOM, num_samples=R, batches=K) draws the K R points of one keyed stream (batch b = global
    restart indices b R ..), improve(COORD_DESCENT) runs them through ONE persistent launch of the lifecycle kernel (n = 64:
    Boolean family, n a multiple of 16) or, where that kernel does not apply (n = 40), as one population -- either way batch b
    must end exactly where the b-th of K serial suggest + improve calls ends, and the variables hold the best of all batches."""
    from qcqp_amd import QCQP, COORD_DESCENT, RANDOM, problems
    from qcqp_amd.form import QCQPForm
    funcs, _, _ = problems.boolean_least_squares(n, 24, seed=6)
    form = QCQPForm.from_arrays(funcs)
    K, R = 3, 40
    q = QCQP(form)
    q.suggest(RANDOM, num_samples=R, batches=K, seed=5)
    f, v = q.improve(COORD_DESCENT, seed=7)
    if n == 64:
        assert q.engine.last_cd_kernel() == 'cd_life_kernel<3,band>'
    assert len(q.batch_results) == K
    q2 = QCQP(form)
    serial = []
    for b in range(K):
        q2.suggest(RANDOM, num_samples=R, seed=5, first_index=b * R)
        fb, vb = q2.improve(COORD_DESCENT, seed=7, first_index=b * R)
        serial.append((fb, vb, q2.best_index, np.array(q2.prob.variables()[0].value).ravel()))
        assert abs(q.batch_results[b]['f'] - fb) <= 1e-11 * (1 + abs(fb)) and abs(q.batch_results[b]['v'] - vb) <= 1e-12, b
        assert q.batch_results[b]['index'] == q2.best_index, b
    from qcqp_amd.dist import better_key
    w = min(range(K), key=lambda b: better_key(serial[b][0], serial[b][1], b))
    assert abs(f - serial[w][0]) <= 1e-11 * (1 + abs(f)) and abs(v - serial[w][1]) <= 1e-12
    assert np.max(np.abs(np.array(q.prob.variables()[0].value).ravel() - serial[w][3])) < 1e-12


def test_circle_packing_two_variables(orc):
    """The fourth example family of the reference (examples/circle_packing.py:6-17): TWO variables -- the centres X (2, N) and
    the radius r -- stacked column-major in the order of prob.variables() (assign_vars / flatten_vars, utilities.py:298-316,
    with the index advanced: SURVEY.md A.3), a linear objective (maximise r), 4 N + 1 linear constraints and N (N - 1) / 2
    sparse indefinite separation constraints that couple five coordinates each.  (1) phase 2 from the reference's golden start
    (G6 circle5: the oracle reproduces the reference's own result on this family, tests/test_oracle_golden.py) through
    Problem(var_sizes=...) and the variables' values: the engine's point against the oracle with the same keyed stream (the
    objective is identically zero in every centre coordinate: each visit draws a uniform point of the feasible set), shapes
    and (f, v) with the maximise sign; (2) a population from suggest(RANDOM) through improve(COORD_DESCENT), restart by restart
    against the oracle; (3) N = 40 (n = 81 > 64: the default dense-constraint path, 941 constraints) by outcome: reported values
    equal to the oracle's evaluation of the returned points, the reference-order mode value for value."""
    from qcqp_amd import QCQP, COORD_DESCENT, RANDOM, Problem, problems
    z = load_golden('g6_cd_circle5')
    N = 5
    funcs, maxi, info = problems.circle_packing(N)
    assert maxi and info['var_sizes'] == [(2, N), (1, 1)]
    prob = Problem(funcs, maximize=True, var_sizes=info['var_sizes'])
    assert rel_eq_forms(prob.qcqp_form, funcs_from_npz(z))
    po = orc.Problem(funcs_from_npz(z))                      # minimise form, as the reference holds it
    q = QCQP(prob)
    Xv, rv = prob.variables()
    x0 = z['X0'][:, 0]
    Xv.value = x0[:2 * N].reshape((2, N), order='F')
    rv.value = x0[2 * N:].reshape((1, 1))
    f, v = q.improve(COORD_DESCENT, phase1=False, seed=11)
    assert q.engine.last_cd_kernel() == 'cd_general_kernel'
    rng = orc.Rng(orc.RNG_KEYED, 11)
    rng.set_restart(0)
    xo, s1, s2 = po.improve_cd(x0, phase1=False, rng=rng)
    assert Xv.value.shape == (2, N) and rv.value.shape == (1, 1)
    xg = np.concatenate([np.ravel(Xv.value, order='F'), np.ravel(rv.value)])
    assert np.max(np.abs(xg - xo)) <= 1e-9 * (1 + np.max(np.abs(xo)))
    assert abs(f - float(rv.value[0, 0])) <= 1e-12 and abs(f + po.eval(0, xo)) <= 1e-9      # maximise: f = r
    assert abs(v - po.max_violation(xo)) <= 1e-9
    # (2) a population of random starts, phase 1 + phase 2
    R = 48
    q.suggest(RANDOM, num_samples=R, seed=3)
    X0 = q.population()
    q.improve(COORD_DESCENT, seed=5, num_iters=40)
    X = q.population()
    for r in range(0, R, 5):
        rng = orc.Rng(orc.RNG_KEYED, 5)
        rng.set_restart(r)
        xo, s1, s2 = po.improve_cd(X0[:, r], num_iters=40, rng=rng)
        assert np.max(np.abs(X[:, r] - xo)) <= 1e-9 * (1 + np.max(np.abs(xo))), r
    best = int(np.argmax(np.where(q.population_v < 1e-2, q.population_f, -np.inf)))
    assert abs(float(rv.value[0, 0]) - X[2 * N, q.best_index]) <= 1e-12 and q.population_v[q.best_index] < 1e-2
    assert q.population_f[q.best_index] >= q.population_f[best] - 1e-4 - 1e-9        # better(): violation bucket first
    # (3) N = 40: the default dense-constraint path by outcome, the reference-order mode value for value
    N2 = 40
    funcs2, _, info2 = problems.circle_packing(N2)
    q2 = QCQP(Problem(funcs2, maximize=True, var_sizes=info2['var_sizes']))
    po2 = orc.Problem(problems.circle_packing(N2, minimize_form=True)[0])
    R2 = 16
    q2.suggest(RANDOM, num_samples=R2, seed=2)
    Y0 = q2.population()
    q2.improve(COORD_DESCENT, seed=6, num_iters=3)
    assert q2.engine.last_cd_kernel().startswith('dense_chain')
    Y = q2.population()
    for r in range(R2):
        assert abs(q2.population_f[r] + po2.eval(0, Y[:, r])) <= 1e-9 * (1 + abs(q2.population_f[r]))
        assert abs(q2.population_v[r] - po2.max_violation(Y[:, r])) <= 1e-9 * (1 + q2.population_v[r])
    # ... and value for value, teacher-forced: the oracle's own states through the unit step of the default path
    # (qcqpmi_cd_dense_block_step, one coordinate visit per step; see tests/test_gpu_scale.py for the statement)
    Rt = 8
    runs = []
    for r in range(Rt):
        rng = orc.Rng(orc.RNG_KEYED, 6)
        rng.set_restart(r)
        runs.append(po2.improve_cd_traced(Y0[:, r], num_iters=3, rng=rng))
    n2 = 2 * N2 + 1
    slack2 = np.array([0.0 if u[5] is None else u[5] for u in runs])
    cur = Y0[:, :Rt].copy()
    worst, cnt = 0.0, 0
    for phase, ti in ((1, 3), (2, 4)):
        trs = [u[ti] for u in runs]
        for v0 in range(max(len(t) for t in trs)):
            t, i = divmod(v0, n2)
            active = [r for r in range(Rt) if len(trs[r]) > v0]
            if not active:
                continue
            q2.engine.upload(cur)
            q2.engine.cd_dense_block_step(phase, t, i // 16, slack=slack2 if phase == 2 else None, seed=6, first_index=0,
                                          coords=(i % 16, i % 16 + 1))
            X1 = q2.engine.download()
            for r in active:
                worst = max(worst, abs(X1[i, r] - trs[r][v0]) / (1 + np.max(np.abs(cur[:, r]))))
                cur[i, r] = trs[r][v0]
                cnt += 1
    print('\ncircle packing N = 40 through the default dense path, teacher-forced: %d visits, worst deviation from the oracle %.1e' % (cnt, worst))
    assert cnt >= 2 * n2 * Rt // 2 and worst < 1e-6


def rel_eq_forms(form, funcs):
    """The QCQPForm a Problem holds equals a raw-array problem (objective first), entry by entry."""
    fs = [form.f0] + list(form.fs)
    if len(fs) != len(funcs):
        return False
    for f, (P, q, r, relop) in zip(fs, funcs):
        Pf = np.asarray(f.P.todense()) if hasattr(f.P, 'todense') else np.asarray(f.P)
        if not (np.array_equal(Pf, np.asarray(P)) and np.array_equal(np.ravel(f.qarray), np.ravel(q)) and f.r == r and f.relop == relop):
            return False
    return True


def test_large_population_takes_the_lifecycle_launch():
    """improve(COORD_DESCENT) on ONE population of 8192 restarts (two generations of the chip's 4096 slots) runs as a lifecycle
    launch (phase 1, gate, phase 2, evaluation in one kernel) -- the same restarts as the separate launches: equal points and
    (f, v), checked against the engine's serial path on the same starts."""
    from qcqp_amd import QCQP, COORD_DESCENT, RANDOM, problems
    from qcqp_amd.form import QCQPForm
    funcs, _, _ = problems.boolean_least_squares(128, 40, seed=8)
    form = QCQPForm.from_arrays(funcs)
    q = QCQP(form)
    q.suggest(RANDOM, num_samples=8192, seed=4)
    X0 = q.population()
    f, v = q.improve(COORD_DESCENT, seed=9)
    assert q.engine.last_cd_kernel() == 'cd_life_kernel<3,band>'
    X = q.population()
    q2 = QCQP(form)
    q2.engine.upload(X0)
    out = q2.engine.cd_run(phase1=True, seed=9)
    assert not q2.engine.last_cd_kernel().startswith('cd_life_kernel')
    assert np.max(np.abs(X - q2.engine.download())) < 1e-12
    idx, fb, vb, xb = q2.engine.select_best(1e-4)
    assert q.best_index == idx and abs(f - fb) <= 1e-11 * (1 + abs(fb)) and abs(v - vb) <= 1e-12
