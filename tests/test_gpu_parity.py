"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle and the golden
vectors captured from the reference.  Run with `-m gpu` on an MI355X."""
import os

import numpy as np
import pytest

from conftest import funcs_from_npz, load_golden, oracle_map

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng_mod():
    from qcqp_amd import engine
    assert engine.device_count() >= 1, 'no HIP device visible'
    return engine


def make(eng_mod, funcs):
    from qcqp_amd.form import QCQPForm
    return eng_mod.Engine(QCQPForm.from_arrays(funcs))


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b)) / (1.0 + np.abs(np.asarray(b))))


@pytest.mark.parametrize('name', ['bls10', 'bls32', 'maxcut12', 'dense16', 'beam10'])
def test_eval_matches_golden(eng_mod, orc, name):
    z = load_golden('g1_' + name)
    e = make(eng_mod, funcs_from_npz(z))
    f0, mv, F = e.eval_batch(z['X'], want_F=True)
    assert rel(F, z['F']) < 1e-12          # tolerance: fp64, summation order differs from SciPy CSR
    assert rel(mv, z['maxviol']) < 1e-12
    assert rel(f0, z['F'][0]) < 1e-12


@pytest.mark.parametrize('name', ['bls10', 'bls32', 'maxcut12', 'dense16', 'beam10'])
def test_onevar_coefficients_match_reference_golden(eng_mod, orc, name):
    """QuadraticFunction.get_onevar_func (utilities.py:99-105) on the device (qcqpmi_onevar_coeffs: the device functions the
    coupled-constraint kernel inlines, csrc/cd_general.h) against the reference's own outputs stored in the G1 fixtures
    (6 points x every function of the problem, tools/gen_golden.py) and, bit for bit, against the oracle's restatement."""
    z = load_golden('g1_' + name)
    e = make(eng_mod, funcs_from_npz(z))
    e.upload(np.ascontiguousarray(z['ov_x'].T))
    T = e.onevar_coeffs(z['ov_k'])
    assert T.shape == z['ov_T'].shape
    assert np.array_equal(T[:, :, 0], z['ov_T'][:, :, 0])                  # t2 = P[k, k]
    # tolerance: the reference's sparse products visit the same terms in another association (measured: <= 3e-15, 85-100 %
    # of the entries bit-identical)
    assert rel(T, z['ov_T']) < 1e-14
    assert (T == z['ov_T']).mean() > 0.8
    # the oracle's restatement at the same points: bit for bit
    prob = orc.Problem(funcs_from_npz(z))
    O = np.array([[prob.onevar_coeffs(j, z['ov_x'][t], int(z['ov_k'][t])) for j in range(T.shape[1])] for t in range(T.shape[0])])
    assert np.array_equal(T, O)


def test_eval_large_vs_oracle(eng_mod, orc):
    from qcqp_amd import problems
    funcs, _, _ = problems.boolean_least_squares(200, 64, seed=5)
    e = make(eng_mod, funcs)
    prob = orc.Problem(funcs)
    X = np.random.RandomState(0).randn(200, 37)
    f0, mv = e.eval_batch(X)
    g0, gv = prob.eval_batch(X)
    assert rel(f0, g0) < 1e-12 and rel(mv, gv) < 1e-13


@pytest.mark.parametrize('generic', [False, True])
@pytest.mark.parametrize('name', ['bls10', 'bls32', 'bls64', 'maxcut12'])
def test_cd_phase2_matches_reference_golden(eng_mod, name, generic):
    """Phase 2 is deterministic: the device trajectory must land on the reference's point.
    generic=True forces the general 4-wave kernel where the pipelined 8-wave one would apply."""
    z = load_golden('g6_cd_' + name)
    e = make(eng_mod, funcs_from_npz(z))
    e.L.qcqpmi_debug_profile(e.h, 2 if generic else 0, None)
    e.upload(z['X0'])
    out = e.cd_run(phase1=False)
    # which kernel ran (a regression of the pipelined kernel's eligibility test must not hide behind its fallbacks):
    # the pipelined kernel takes n = 48 ... 1024 in multiples of 16 (>= 3 blocks), n = 32 goes to its round-1
    # predecessor, sizes that are not a multiple of 16 (padding coordinates form a second constraint class) to the
    # general kernel
    want = 'cd_phase2_kernel' if generic else {'bls10': 'cd_phase2_kernel', 'bls32': 'cd_phase2_rs_kernel',
                                               'bls64': 'cd_phase2_q_kernel', 'maxcut12': 'cd_phase2_kernel'}[name]
    assert e.last_cd_kernel() == want, (e.last_cd_kernel(), want)
    X = e.download()
    assert rel(X, z['p2_x']) < 1e-9
    assert rel(out['f0'], z['p2_fv'][:, 0]) < 1e-9
    assert rel(out['maxviol'], z['p2_fv'][:, 1]) < 1e-9


@pytest.mark.parametrize('generic', [False, True])
@pytest.mark.parametrize('name,n,m_rows', [('bls', 96, 40), ('bls', 250, 100), ('maxcut', 130, 0), ('maxcut', 128, 0)])
def test_cd_full_driver_vs_oracle_keyed(eng_mod, orc, name, n, m_rows, generic):
    """Phase 1 draws from the keyed Philox stream: oracle (ORC_RNG_KEYED) and GPU consume the same
    draws, so whole improve_coord_descent trajectories are comparable."""
    from qcqp_amd import problems
    funcs = (problems.boolean_least_squares(n, m_rows, seed=2)[0] if name == 'bls'
             else problems.maxcut(n, 0.5, seed=3, weighted=True)[0])
    e = make(eng_mod, funcs)
    e.L.qcqpmi_debug_profile(e.h, 2 if generic else 0, None)
    prob = orc.Problem(funcs)
    R, seed, first = 21, 1234, 7
    X0 = np.random.RandomState(1).randn(n, R)
    e.upload(X0)
    out = e.cd_run(phase1=True, num_iters=50, seed=seed, first_index=first)
    # (bls, 96): the pipelined kernel; (maxcut, 128): zero diagonal, multiple of 16 -> its predecessor; 250 / 130: general
    want = 'cd_phase2_kernel' if generic else {('bls', 96): 'cd_phase2_q_kernel', ('maxcut', 128): 'cd_phase2_rs_kernel'}.get((name, n), 'cd_phase2_kernel')
    assert e.last_cd_kernel() == want, (e.last_cd_kernel(), want)
    X = e.download()
    for r in range(R):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(first + r)
        x, s1, s2 = prob.improve_cd(X0[:, r], num_iters=50, rng=rng)
        assert rel(X[:, r], x) < 1e-9, r
        assert out['sweeps1'][r] == s1[0]
        assert out['visits2'][r] == s2[1] and out['accepted2'][r] == s2[2]
        assert abs(out['f0'][r] - prob.eval(0, x)) <= 1e-9 * (1 + abs(out['f0'][r]))


def test_randn_matches_oracle_stream(eng_mod, orc):
    from qcqp_amd import problems
    funcs, _, _ = problems.boolean_least_squares(33, 10, seed=1)
    e = make(eng_mod, funcs)
    e.randn(19, seed=99, first_index=1000)
    X = e.download()
    ref = orc.keyed_normal_matrix(99, 33, 19, first_index=1000)
    assert np.max(np.abs(X - ref)) < 1e-13   # device libm vs glibc: last-ulp differences allowed
    assert abs(X.mean()) < 0.2 and 0.8 < X.std() < 1.2


def test_sdr_sample_affine(eng_mod, orc):
    z = load_golden('g9_sdr_bls10')
    funcs = funcs_from_npz(z)
    e = make(eng_mod, funcs)
    n = 10
    mu, Sigma = orc.sdr_mu_sigma(z['X'], compat=False)
    F = np.linalg.cholesky(Sigma)
    Xi = np.random.RandomState(3).randn(n, 40)
    e.sdr_sample(mu, F, 40, Xi=Xi)
    X = e.download()
    assert rel(X, mu[:, None] + F.dot(Xi)) < 1e-13
    f0, mv = e.eval()
    g0, gv = orc.Problem(funcs).eval_batch(X)
    assert rel(f0, g0) < 1e-12 and rel(mv, gv) < 1e-12
    # device normals: same stream as the oracle's keyed normals
    e.sdr_sample(mu, F, 24, seed=5, first_index=3)
    Xd = e.download()
    Xi2 = orc.keyed_normal_matrix(5, n, 24, first_index=3)
    assert rel(Xd, mu[:, None] + F.dot(Xi2)) < 1e-12


def test_select_best_ordering(eng_mod, orc):
    z = load_golden('g1_bls10')
    funcs = funcs_from_npz(z)
    e = make(eng_mod, funcs)
    e.upload(z['X'])
    idx, f, v, x = e.select_best(1e-4)
    prob = orc.Problem(funcs)
    best = 0
    for s in range(1, z['X'].shape[1]):   # fold with QCQPForm.better, ties keep the earlier index
        if prob.better(z['X'][:, s], z['X'][:, best]) == 1:
            v1 = int(prob.max_violation(z['X'][:, s]) / 1e-4), prob.eval(0, z['X'][:, s])
            v2 = int(prob.max_violation(z['X'][:, best]) / 1e-4), prob.eval(0, z['X'][:, best])
            if v1 != v2:
                best = s
    assert idx == best
    assert np.array_equal(x, z['X'][:, idx])


@pytest.mark.parametrize('name', ['dense16', 'dense32'])
def test_cd_coupled_constraints_phase2_matches_reference_golden(eng_mod, name):
    """Dense constraints that couple all coordinates (general kernel): phase 2 lands on the
    reference's own points (golden G6)."""
    z = load_golden('g6_cd_' + name)
    e = make(eng_mod, funcs_from_npz(z))
    assert not e.separable
    e.upload(z['X0'])
    out = e.cd_run(phase1=False)
    assert e.last_cd_kernel() == 'cd_general_kernel'
    X = e.download()
    assert rel(X, z['p2_x']) < 1e-8
    assert rel(out['f0'], z['p2_fv'][:, 0]) < 1e-8
    assert np.max(np.abs(out['maxviol'] - z['p2_fv'][:, 1])) < 1e-8


@pytest.mark.parametrize('family', ['dense', 'beam'])
def test_cd_coupled_constraints_full_driver_vs_oracle(eng_mod, orc, family):
    from qcqp_amd import problems
    if family == 'dense':
        funcs, _, _ = problems.dense_indefinite(24, 6, seed=11)
        X0 = 0.3 * np.random.RandomState(1).randn(24, 10)
    else:
        funcs, _, _ = problems.beamforming(8, 3, 2, seed=4)
        X0 = 2.0 * np.random.RandomState(1).randn(16, 10)
    e = make(eng_mod, funcs)
    prob = orc.Problem(funcs)
    seed, first, iters = 99, 5, 40
    e.upload(X0)
    out = e.cd_run(phase1=True, num_iters=iters, seed=seed, first_index=first)
    X = e.download()
    for r in range(X0.shape[1]):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(first + r)
        x, s1, s2 = prob.improve_cd(X0[:, r], num_iters=iters, rng=rng)
        assert rel(X[:, r], x) < 1e-8, (r, np.max(np.abs(X[:, r] - x)))
        assert out['sweeps1'][r] == s1[0] and out['visits2'][r] == s2[1] and out['accepted2'][r] == s2[2], r


# ----------------------------------------------------------------------------- full-size config
def test_full_size_headline_config(eng_mod, orc):
    """BASELINE.json configs[1]: Boolean least squares n=1024, m=256, 4096 restarts.
    * three restarts are checked against the oracle trajectory (the oracle needs ~15 s each);
    * size-independent properties on all 4096: sharding invariance (a slice of the restarts run
      alone with the matching global index offset reproduces the big run bit for bit),
      idempotence (phase 2 from a converged point moves nothing), every restart feasible within
      the slack phase 1 leaves, and the reported objective equals a fresh evaluation."""
    from qcqp_amd import problems
    n, R, seed = 1024, 4096, 2024
    funcs, _, _ = problems.boolean_least_squares(n, 256, seed=1)
    e = make(eng_mod, funcs)
    e.randn(R, seed=seed)
    X0 = e.download()
    out = e.cd_run(seed=seed)
    assert e.last_cd_kernel() == 'cd_phase2_q_kernel'      # the headline kernel, not one of its fallbacks
    X = e.download()
    ran = out['ran_phase2'].astype(bool)
    # a few restarts get stuck in phase 1 exactly like in the reference: a coordinate whose
    # violation is within one bisection tolerance above viol_tol is never moved (qcqp.py:122-132),
    # the gate of qcqp.py:189 then skips phase 2 for that restart
    assert ran.mean() > 0.9
    assert np.all(out['maxviol'][~ran] >= 1e-2) and np.all(out['visits2'][~ran] == 0)
    # phase 2 keeps every constraint within the slack phase 1 left (< viol_tol = 1e-2)
    assert out['maxviol'][ran].max() < 1e-2
    assert np.all(np.abs(X[:, ran] ** 2 - 1.0) < 1e-2)
    f0, mv = e.eval()
    # the pipelined kernel reports the objective it tracked through every accepted move (no evaluation pass
    # afterwards): equal to a fresh evaluation up to the rounding of ~1e4 updates; violations are recomputed
    # from the final tile with the evaluation kernel's own expression: bit-identical
    assert np.max(np.abs(f0 - out['f0']) / (1.0 + np.abs(f0))) < 1e-11 and np.array_equal(mv, out['maxviol'])
    # oracle trajectories
    prob = orc.Problem(funcs)
    stuck = int(np.flatnonzero(~ran)[0]) if (~ran).any() else 4095
    def oracle_restart(r):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(r)
        return prob.improve_cd(X0[:, r], num_iters=(1000 if ran[r] else 5), rng=rng)
    picks = (0, 1777, stuck)
    for r, (x, s1, s2) in zip(picks, oracle_map(oracle_restart, picks)):      # ~15 s of oracle each: side by side
        assert rel(X[:, r], x) < 1e-9, r
        assert out['visits2'][r] == s2[1] and out['accepted2'][r] == s2[2]
        assert abs(out['f0'][r] - prob.eval(0, x)) <= 1e-6 * abs(out['f0'][r])   # north-star tolerance
    # sharding invariance: restarts [2048, 2048+512) alone, as a second "rank" would run them
    e.randn(512, seed=seed, first_index=2048)
    out2 = e.cd_run(seed=seed, first_index=2048)
    X2 = e.download()
    assert np.array_equal(X2, X[:, 2048:2560])
    assert np.array_equal(out2['f0'], out['f0'][2048:2560])
    # idempotence of phase 2
    e.upload(X[:, :256])
    out3 = e.cd_run(phase1=False)
    assert np.array_equal(e.download(), X[:, :256])
    assert out3['accepted2'].sum() == 0 and np.all(out3['visits2'][ran[:256]] == n)
    # best-of-population rule


# ------------------------------------------------------------------------------------- ADMM
@pytest.mark.parametrize('name', ['beam10', 'beam40', 'bls10', 'dense16'])
def test_admm_matches_reference_golden(eng_mod, orc, name):
    """improve_admm on the GPU (eigenbasis formulation: the engine's own fp64 MFMA GEMM for the two products per
    iteration + the hand-written secular kernel; no vendor BLAS) against the reference's own result (golden G8), fed the
    reference's eigenpairs."""
    z = load_golden('g8_admm_' + name)
    funcs = funcs_from_npz(z)
    e = make(eng_mod, funcs)
    n = int(z['n'])
    m = len(funcs) - 1
    e.admm_set_eig(z['lmb'], z['Q'])
    rho = float(z['rho'])
    P0 = np.asarray(funcs[0][0])
    Minv = np.linalg.inv(2. * (P0 + rho * m * np.eye(n)))
    iters = int(z['iters'])
    # several copies of the same start: every restart must follow the reference
    e.upload(np.stack([z['x0']] * 5, axis=1))
    out = e.admm_run(rho, Minv, phase1=True, num_iters=iters)
    X = e.download()
    for r in range(5):
        assert rel(X[:, r], z['xa']) < 1e-6, (name, r)     # north-star tolerance (bisection tol is 1e-6)
    assert rel(out['f0'], np.full(5, z['fva'][0])) < 1e-6
    assert np.max(np.abs(out['maxviol'] - z['fva'][1])) < 1e-6 * (1 + abs(z['fva'][1]))


def test_admm_population_vs_oracle(eng_mod, orc):
    from qcqp_amd import problems
    funcs, _, _ = problems.beamforming(12, 4, 3, seed=2)
    e = make(eng_mod, funcs)
    prob = orc.Problem(funcs)
    n, m = prob.n, prob.m
    lm, Q = prob.eig()
    e.admm_set_eig(lm, Q)
    rho = float(np.sqrt(m))
    P0 = np.asarray(funcs[0][0])
    Minv = np.linalg.inv(2. * (P0 + rho * m * np.eye(n)))
    R = 9
    X0 = np.random.RandomState(4).randn(n, R)
    e.upload(X0)
    out = e.admm_run(rho, Minv, phase1=True, num_iters=80)
    X = e.download()
    for r in range(R):
        xa = prob.improve_admm(X0[:, r], num_iters=80, rho=rho)
        assert rel(X[:, r], xa) < 1e-6, r
        assert abs(out['f0'][r] - prob.eval(0, xa)) <= 1e-6 * (1 + abs(out['f0'][r]))


# --------------------------------------------------------------- general separable constraints
def mixed_separable(n, per_coord, seed, objective='psd'):
    """Every coordinate carries `per_coord` constraints of random kinds that all leave a common
    feasible region around +-[0.6, 1.6]: box (linear <=), disc (x^2 <= c), annulus (-x^2 <= -c),
    equality x^2 == c with slack.  Several constraint classes, two-interval sets, infinite end points."""
    import scipy.sparse as sp
    rs = np.random.RandomState(seed)
    G = rs.randn(n, n)
    if objective == 'psd':
        P0 = G.dot(G.T) / n + 0.5 * np.eye(n)
    elif objective == 'indef':
        P0 = (G + G.T) / 2.
        P0[np.arange(0, n, 3), np.arange(0, n, 3)] = 0.0          # zero, positive and negative curvature
    else:
        P0 = (G + G.T) / 2.
        np.fill_diagonal(P0, 0.0)
    funcs = [(P0, rs.randn(n), 0.3, None)]
    kinds = ['box_hi', 'box_lo', 'annulus']
    for i in range(n):
        # the first constraint keeps the feasible set bounded (the reference raises OverflowError in
        # phase 1 on unbounded sets, see test_cd_unbounded_set_raises_like_reference)
        chosen = [rs.choice(['disc', 'eq'])] + [kinds[c] for c in rs.choice(len(kinds), size=per_coord - 1, replace=False)]
        for kind in chosen:
            P = np.zeros((n, n)); q = np.zeros(n)
            if kind == 'box_hi':
                q[i] = 1.0; r = -(1.7 + 0.1 * rs.rand()); P = sp.csr_matrix((n, n))
            elif kind == 'box_lo':
                q[i] = -1.0; r = -(1.8 + 0.1 * rs.rand()); P = sp.csr_matrix((n, n))
            elif kind == 'disc':
                P = sp.csr_matrix(([1.0], ([i], [i])), shape=(n, n)); r = -(2.5 + rs.rand())
            elif kind == 'eq':
                P = sp.csr_matrix(([1.0], ([i], [i])), shape=(n, n)); r = -(1.0 + 0.5 * rs.rand())
                funcs.append((P, q, r, '=='))
                continue
            else:
                P = sp.csr_matrix(([-1.0], ([i], [i])), shape=(n, n)); r = 0.3 + 0.1 * rs.rand()
            funcs.append((P, q, r, '<='))
    return funcs


@pytest.mark.parametrize('per_coord,objective', [(1, 'psd'), (1, 'indef'), (2, 'psd'), (3, 'indef'), (2, 'zero')])
def test_cd_general_separable_vs_oracle(eng_mod, orc, per_coord, objective):
    """Exercises the general phase-1/phase-2 kernels (several constraints per coordinate, many
    constraint classes, unbounded intervals, all curvature signs) against the oracle."""
    n, R, seed = 40, 19, 77
    funcs = mixed_separable(n, per_coord, seed=5 + per_coord, objective=objective)
    e = make(eng_mod, funcs)
    assert e.separable
    prob = orc.Problem(funcs)
    X0 = 1.2 * np.random.RandomState(2).randn(n, R)
    e.upload(X0)
    try:
        out = e.cd_run(phase1=True, num_iters=60, seed=seed, first_index=3)
        gpu_err = None
        assert e.last_cd_kernel() == 'cd_phase2_kernel'
    except eng_mod.EngineError as err:
        gpu_err = str(err)
    X = e.download()
    for r in range(R):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(3 + r)
        x, s1, s2 = prob.improve_cd(X0[:, r], num_iters=60, rng=rng)
        assert gpu_err is None, gpu_err
        assert rel(X[:, r], x) < 1e-9, (r, np.max(np.abs(X[:, r] - x)))
        assert out['visits2'][r] == s2[1] and out['accepted2'][r] == s2[2], r


def test_cd_unbounded_set_raises_like_reference(eng_mod, orc):
    """A coordinate whose only constraint is a one-sided bound has an unbounded feasible interval:
    the reference's phase 1 dies in np.random.uniform (OverflowError); the engine reports it."""
    import scipy.sparse as sp
    n = 8
    funcs = [(np.eye(n), np.zeros(n), 0.0, None)]
    for i in range(n):
        q = np.zeros(n); q[i] = 1.0
        funcs.append((sp.csr_matrix((n, n)), q, -1.0, '<='))
    e = make(eng_mod, funcs)
    prob = orc.Problem(funcs)
    x0 = 3.0 + np.abs(np.random.RandomState(0).randn(n))
    with pytest.raises(RuntimeError):
        prob.improve_cd(x0, rng=orc.Rng(orc.RNG_KEYED, 1))
    e.upload(x0)
    with pytest.raises(eng_mod.EngineError) as ei:
        e.cd_run(seed=1)
    assert 'OverflowError' in str(ei.value)


def test_cd_coupled_constraints_tracked_mode_quality(eng_mod, orc):
    """n > 64: t0 comes from the incrementally tracked f_k(x).  Coordinate descent on this family is
    chaotic in the reference itself (active constraints make the interval end points roots of
    near-degenerate quadratics: a 1e-15 difference in f_k moves them by 1e-8 and soon flips a branch;
    with the reference's own arithmetic for t0 -- n <= 64 -- the GPU reproduces the oracle bit for
    bit, see test_cd_coupled_constraints_full_driver_vs_oracle).  Here the two populations are compared
    statistically: feasibility rate and the objective distribution of the feasible restarts."""
    from qcqp_amd import problems
    funcs, _, _ = problems.beamforming(40, 4, 2, seed=6)
    n = 80
    e = make(eng_mod, funcs)
    prob = orc.Problem(funcs)
    R, seed, iters = 64, 3, 15
    X0 = 2.0 * np.random.RandomState(7).randn(n, R)
    e.upload(X0)
    out = e.cd_run(phase1=True, num_iters=iters, seed=seed)
    fo, vo = [], []
    for r in range(R):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(r)
        x, s1, s2 = prob.improve_cd(X0[:, r], num_iters=iters, rng=rng)
        fo.append(prob.eval(0, x)); vo.append(prob.max_violation(x))
    fo, vo = np.array(fo), np.array(vo)
    okg, oko = out['maxviol'] < 1e-2, vo < 1e-2
    print('feasible gpu/oracle', okg.mean(), oko.mean(), 'median f', np.median(out['f0'][okg]), np.median(fo[oko]),
          'quartiles', np.percentile(out['f0'][okg], [25, 75]), np.percentile(fo[oko], [25, 75]))
    assert abs(okg.mean() - oko.mean()) <= 0.15
    assert abs(np.median(out['f0'][okg]) - np.median(fo[oko])) <= 0.3 * np.median(fo[oko])
    # every point the engine calls feasible IS feasible for the oracle's evaluation
    X = e.download()
    for r in np.flatnonzero(okg)[:8]:
        assert abs(prob.max_violation(X[:, r]) - out['maxviol'][r]) < 1e-9
        assert abs(prob.eval(0, X[:, r]) - out['f0'][r]) <= 1e-9 * (1 + abs(out['f0'][r]))


@pytest.mark.parametrize('per_coord,n', [(5, 24), (8, 24), (6, 80)])
def test_cd_more_than_four_constraints_per_coordinate(eng_mod, orc, per_coord, n):
    """utilities.py:241-255 takes any number of constraints on a coordinate; the per-lane solver of the separable
    kernels holds four.  Round 2 refused such problems (QCQPMI_EUNSUPPORTED); now they are routed to the paths for
    general constraints (n <= 64: cd_general_kernel in the reference's arithmetic, trajectories equal to the oracle's;
    above: the dense path, compared like every run of that path -- reported values exact, same feasibility class)."""
    import scipy.sparse as sp
    rs = np.random.RandomState(100 + per_coord)
    G = rs.randn(n, n)
    funcs = [(G.dot(G.T) / n + 0.5 * np.eye(n), rs.randn(n), 0.3, None)]
    for i in range(n):
        E = sp.csr_matrix(([1.0], ([i], [i])), shape=(n, n))
        funcs.append((E, np.zeros(n), -(2.6 + 0.3 * rs.rand()), '<='))           # disc: keeps the set bounded
        for kind in rs.choice(3, size=per_coord - 1):
            q = np.zeros(n)
            if kind == 0:
                q[i] = 1.0; funcs.append((sp.csr_matrix((n, n)), q, -(1.5 + 0.1 * rs.rand()), '<='))      # x <= 1.5..1.6
            elif kind == 1:
                q[i] = -1.0; funcs.append((sp.csr_matrix((n, n)), q, -(1.5 + 0.1 * rs.rand()), '<='))     # x >= -1.6..-1.5
            else:
                funcs.append((-E, q, 0.3 + 0.1 * rs.rand(), '<='))                                         # x^2 >= 0.3..0.4
    e = make(eng_mod, funcs)
    assert not e.separable            # more than four per coordinate: handled as general constraints
    prob = orc.Problem(funcs)
    R, seed, first, iters = 12, 31, 2, 40 if n <= 64 else 6
    X0 = 1.3 * np.random.RandomState(6).randn(n, R)
    e.upload(X0)
    out = e.cd_run(phase1=True, num_iters=iters, seed=seed, first_index=first)
    assert e.last_cd_kernel() == ('cd_general_kernel' if n <= 64 else 'dense_chain_mw_kernel')
    X = e.download()
    same = 0
    for r in range(R):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(first + r)
        x, s1, s2 = prob.improve_cd(X0[:, r], num_iters=iters, rng=rng)
        assert abs(prob.eval(0, X[:, r]) - out['f0'][r]) <= 1e-9 * (1 + abs(out['f0'][r]))
        assert abs(prob.max_violation(X[:, r]) - out['maxviol'][r]) < 1e-9
        if n <= 64:
            assert rel(X[:, r], x) < 1e-9, (r, np.max(np.abs(X[:, r] - x)))
            assert out['sweeps1'][r] == s1[0] and out['visits2'][r] == s2[1] and out['accepted2'][r] == s2[2], r
        else:
            same += rel(X[:, r], x) < 1e-6
            assert (out['maxviol'][r] < 1e-2) == (prob.max_violation(x) < 1e-2), r
    if n > 64:
        print('\n%d constraints per coordinate, n = %d through the dense path: %d of %d restarts on the oracle trajectory' % (per_coord, n, same, R))
        if prob.m > 240:
            return      # the reference-order kernel keeps a coefficient table of (m + 1) x 16 x 4 doubles in LDS: m <= 240
        e.cd_reference_order(True)
        e.upload(X0)
        e.cd_run(phase1=True, num_iters=iters, seed=seed, first_index=first)
        Xr = e.download()
        for r in range(R):
            rng = orc.Rng(orc.RNG_KEYED, seed)
            rng.set_restart(first + r)
            x, s1, s2 = prob.improve_cd(X0[:, r], num_iters=iters, rng=rng)
            assert rel(Xr[:, r], x) < 1e-9, (r, np.max(np.abs(Xr[:, r] - x)))


# ------------------------------------------------------- dense-constraint path (matrix cores)
DENSE_PATH = 32 << 4   # qcqpmi_debug_profile switch: take the dense path on small problems too (default: n > 64)


def test_dense_path_eval_matches_oracle(eng_mod, orc):
    """All m+1 quadratic forms on the matrix cores (dense_products_kernel<1>) against the oracle's
    row-by-row evaluation; ragged sizes (n, m+1 and R not multiples of the tile sizes)."""
    from qcqp_amd import problems
    for (n, m, R) in [(100, 7, 37), (70, 13, 300)]:
        funcs, _, _ = problems.dense_indefinite(n, m, seed=3)
        e = make(eng_mod, funcs)
        prob = orc.Problem(funcs)
        X = np.random.RandomState(n).randn(n, R)
        f0, mv, F = e.eval_batch(X, want_F=True)
        g0, gv = prob.eval_batch(X)
        assert rel(f0, g0) < 1e-12 and rel(mv, gv) < 1e-12
        for r in (0, R - 1):
            for k in range(m + 1):
                assert abs(F[k, r] - prob.eval(k, X[:, r])) <= 1e-11 * (1 + abs(F[k, r]))


@pytest.mark.parametrize('name', ['dense16', 'dense32'])
def test_dense_path_phase2_matches_reference_golden(eng_mod, name):
    """Blocked products + bounds/gap sweep: same points as the reference (golden G6) up to rounding
    (MFMA summation order, tracked f_k)."""
    z = load_golden('g6_cd_' + name)
    e = make(eng_mod, funcs_from_npz(z))
    e.L.qcqpmi_debug_profile(e.h, DENSE_PATH, None)
    e.upload(z['X0'])
    out = e.cd_run(phase1=False)
    assert e.last_cd_kernel() == 'dense_chain_mw_kernel'
    X = e.download()
    assert rel(X, z['p2_x']) < 1e-6
    assert rel(out['f0'], z['p2_fv'][:, 0]) < 1e-6
    assert np.max(np.abs(out['maxviol'] - z['p2_fv'][:, 1])) < 1e-6


@pytest.mark.parametrize('family', ['dense', 'beam'])
def test_dense_path_full_driver_vs_general_kernel(eng_mod, orc, family):
    """Phase 1 + gate + phase 2 through the dense path against the oracle (same keyed random
    stream).  The dense family follows the oracle's trajectory to rounding; the beamforming family is
    chaotic (see test_cd_coupled_constraints_tracked_mode_quality) and is compared by outcome."""
    from qcqp_amd import problems
    if family == 'dense':
        funcs, _, _ = problems.dense_indefinite(24, 6, seed=11)
        X0 = 0.3 * np.random.RandomState(1).randn(24, 10)
    else:
        funcs, _, _ = problems.beamforming(8, 3, 2, seed=4)
        X0 = 2.0 * np.random.RandomState(1).randn(16, 10)
    e = make(eng_mod, funcs)
    e.L.qcqpmi_debug_profile(e.h, DENSE_PATH, None)
    prob = orc.Problem(funcs)
    seed, first, iters = 99, 5, 40
    e.upload(X0)
    out = e.cd_run(phase1=True, num_iters=iters, seed=seed, first_index=first)
    X = e.download()
    close = 0
    for r in range(X0.shape[1]):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(first + r)
        x, s1, s2 = prob.improve_cd(X0[:, r], num_iters=iters, rng=rng)
        # reported values are those of the returned point
        assert abs(prob.max_violation(X[:, r]) - out['maxviol'][r]) < 1e-9
        assert abs(prob.eval(0, X[:, r]) - out['f0'][r]) <= 1e-9 * (1 + abs(out['f0'][r]))
        if rel(X[:, r], x) < 1e-6:
            close += 1
            assert out['sweeps1'][r] == s1[0]
        # never worse in feasibility class than the oracle's run of the same restart
        assert (out['maxviol'][r] < 1e-2) == (prob.max_violation(x) < 1e-2), r
    print(family, 'restarts on the oracle trajectory:', close, 'of', X0.shape[1])
    if family == 'dense':
        assert close == X0.shape[1]


def test_dense_path_default_dispatch_vs_oracle(eng_mod, orc):
    """n > 64 takes the dense path without any switch: dense indefinite family, phase 1 + phase 2
    against the oracle with the same keyed stream (trajectories agree to rounding: this family is not
    chaotic), ragged sizes (n, m + 1, R not multiples of 16 / 8 / 16)."""
    from qcqp_amd import problems
    n, m, R = 100, 11, 5
    funcs, _, _ = problems.dense_indefinite(n, m, seed=5)
    e = make(eng_mod, funcs)
    prob = orc.Problem(funcs)
    seed, first, iters = 7, 3, 6
    X0 = 1.5 * np.random.RandomState(2).randn(n, R)
    e.upload(X0)
    out = e.cd_run(phase1=True, num_iters=iters, seed=seed, first_index=first)
    assert e.last_cd_kernel() == 'dense_chain_mw_kernel'
    X = e.download()
    exact = 0
    for r in range(R):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(first + r)
        x, s1, s2 = prob.improve_cd(X0[:, r], num_iters=iters, rng=rng)
        d = np.max(np.abs(X[:, r] - x))
        # a move of size tol +- rounding may be accepted on one side and rejected on the other
        # (qcqp.py:168): such restarts differ by O(tol) in a few coordinates, never more
        assert d < 1e-3, (r, d)
        exact += d < 1e-6 * (1 + np.max(np.abs(x)))
        fo = prob.eval(0, x)
        assert abs(out['f0'][r] - fo) <= 1e-4 * (1 + abs(fo)), r
        assert (out['maxviol'][r] < 1e-2) == (prob.max_violation(x) < 1e-2), r
        assert out['sweeps1'][r] == s1[0], r
        assert abs(prob.eval(0, X[:, r]) - out['f0'][r]) <= 1e-9 * (1 + abs(out['f0'][r]))
        assert abs(prob.max_violation(X[:, r]) - out['maxviol'][r]) < 1e-9
    print('restarts identical to rounding:', exact, 'of', R)
    assert exact >= R // 2


def test_generated_functions_match_oracle_stream(eng_mod, orc):
    """qcqpmi_set_quad_generated: matrices synthesised on the device (cfg5 at full size never exists
    on the host) are the ones the oracle's keyed generator defines -- checked through the function
    values of random points, and a CD run on the generated problem against the oracle on the
    materialised one."""
    from qcqp_amd import problems
    n, m, R = 40, 6, 9
    form = problems.dense_indefinite_generated(n, m, seed=21)
    funcs = problems.materialise_generated(form, orc.keyed_normal)
    e = eng_mod.Engine(form)
    assert not e.separable
    prob = orc.Problem(funcs)
    X = 0.7 * np.random.RandomState(3).randn(n, R)
    f0, mv, F = e.eval_batch(X, want_F=True)
    for r in range(R):
        for k in range(m + 1):
            assert abs(F[k, r] - prob.eval(k, X[:, r])) <= 1e-11 * (1 + abs(F[k, r])), (k, r)
    g0, gv = prob.eval_batch(X)
    assert rel(f0, g0) < 1e-11 and rel(mv, gv) < 1e-11
    out = e.cd_run(phase1=True, num_iters=8, seed=5)
    Xg = e.download()
    close = 0
    for r in range(R):
        rng = orc.Rng(orc.RNG_KEYED, 5)
        rng.set_restart(r)
        x, s1, s2 = prob.improve_cd(X[:, r], num_iters=8, rng=rng)
        assert np.max(np.abs(Xg[:, r] - x)) < 1e-3
        close += np.max(np.abs(Xg[:, r] - x)) < 1e-6
        assert abs(prob.eval(0, Xg[:, r]) - out['f0'][r]) <= 1e-9 * (1 + abs(out['f0'][r]))
    assert close >= R // 2


@pytest.mark.parametrize('n', [20, 32])
def test_cd_phase2_exact_ties_take_the_reference_path(eng_mod, orc, n):
    """The fast path of the pipelined kernel hands every decision close to a tie to a loop that follows
    the reference's arithmetic literally.  Objective x'x with x_i^2 == 1: the vertex of every scalar
    problem is exactly 0, the midpoint between the two feasible intervals -- EVERY visit is a tie, the
    reference breaks it with np.random.choice (utilities.py:283-288), here with the keyed stream the
    oracle shares.  Both block shapes (n a multiple of 16 or not)."""
    funcs = [(np.eye(n), np.zeros(n), 0.0, None)]
    for i in range(n):
        P = np.zeros((n, n)); P[i, i] = 1.0
        funcs.append((P, np.zeros(n), -1.0, '=='))
    e = make(eng_mod, funcs)
    assert e.separable
    prob = orc.Problem(funcs)
    R, seed, first = 7, 77, 2
    rs = np.random.RandomState(n)
    X0 = np.sign(rs.randn(n, R)) * (1.0 + 2e-3 * rs.rand(n, R))
    e.upload(X0)
    out = e.cd_run(phase1=False, num_iters=30, seed=seed, first_index=first)
    X = e.download()
    for r in range(R):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(first + r)
        x, s1, s2 = prob.improve_cd(X0[:, r], num_iters=30, phase1=False, rng=rng)
        assert rel(X[:, r], x) < 1e-12, (r, np.max(np.abs(X[:, r] - x)))
        assert out['visits2'][r] == s2[1] and out['accepted2'][r] == s2[2], r


@pytest.mark.parametrize('n,m,R,iters,p1', [(8, 3, 1, 5, True), (16, 2, 3, 5, True), (17, 4, 16, 8, True),
                                            (33, 5, 17, 6, False), (40, 3, 33, 0, True), (24, 6, 5, 1, True),
                                            (65, 2, 48, 3, True)])
def test_dense_path_edge_shapes(eng_mod, orc, n, m, R, iters, p1):
    """Ragged and degenerate shapes through the dense path (one block only, n not a multiple of 16, a single
    restart, R not a multiple of 16, zero or one sweep, phase 1 skipped): points against the oracle's
    trajectories, reported values against fresh oracle evaluations."""
    from qcqp_amd import problems
    funcs, _, _ = problems.dense_indefinite(n, m, seed=n + m)
    e = make(eng_mod, funcs)
    e.L.qcqpmi_debug_profile(e.h, DENSE_PATH, None)
    prob = orc.Problem(funcs)
    X0 = 0.5 * np.random.RandomState(n).randn(n, R)
    e.upload(X0)
    out = e.cd_run(phase1=p1, num_iters=iters, seed=9, first_index=4)
    X = e.download()
    for r in range(R):
        rng = orc.Rng(orc.RNG_KEYED, 9)
        rng.set_restart(4 + r)
        x, s1, s2 = prob.improve_cd(X0[:, r], num_iters=iters, phase1=p1, rng=rng)
        assert np.max(np.abs(X[:, r] - x)) < 1e-3, r             # O(tol) at most (threshold flips), usually 1e-12
        assert abs(prob.eval(0, X[:, r]) - out['f0'][r]) <= 1e-9 * (1 + abs(out['f0'][r]))
        assert abs(prob.max_violation(X[:, r]) - out['maxviol'][r]) < 1e-9


# ------------------------------------------------------- restart-level scheduling of phase 2 (round 3)
def test_cd_queue_kernel_vs_tile_bound(eng_mod, orc):
    """cd_phase2_qs_kernel (16 slots per workgroup, refilled from a device-side queue at sweep boundaries) against the
    tile-bound cd_phase2_q_kernel and the oracle: 2000 restarts of Boolean least squares n = 256 (125 tiles -> with the
    queue forced on, 125 workgroups that each see several refills... and a second run on 40 workgroups' worth of
    restarts so that every slot is refilled many times is covered by R = 2000 / few CUs not being controllable here; the
    bench covers many generations).  Asserted: same points (1e-12; in practice bit-identical), identical visit / accept /
    sweep counters, tracked objective to 1e-12; a repeat is bit-identical (results do not depend on the scheduling); a
    slice of the restarts run alone with its global index offset is bit-identical (sharding invariance); three restarts
    follow the oracle trajectory."""
    from qcqp_amd import problems
    n, R, seed = 256, 2000, 77
    funcs, _, _ = problems.boolean_least_squares(n, 64, seed=4)
    e = make(eng_mod, funcs)
    res = []
    for mode in (0, 1, 1):
        e.cd_queue(mode)
        e.randn(R, seed=seed)
        X0 = e.download()
        out = e.cd_run(seed=seed)
        assert e.last_cd_kernel() == ('cd_phase2_qs_kernel' if mode else 'cd_phase2_q_kernel')
        res.append((e.download(), out))
    (Xt, ot), (Xq, oq), (Xq2, oq2) = res
    assert np.array_equal(Xq, Xq2) and np.array_equal(oq['f0'], oq2['f0']) and np.array_equal(oq['maxviol'], oq2['maxviol'])
    assert rel(Xq, Xt) < 1e-12
    for key in ('sweeps1', 'sweeps2', 'visits2', 'accepted2', 'ran_phase2', 'status1', 'status2'):
        assert np.array_equal(oq[key], ot[key]), key
    assert rel(oq['f0'], ot['f0']) < 1e-12 and np.array_equal(oq['maxviol'], ot['maxviol'])
    f0, mv = e.eval()
    assert rel(oq['f0'], f0) < 1e-11 and np.array_equal(mv, oq['maxviol'])
    # sharding invariance in queue mode
    e.randn(512, seed=seed, first_index=512)
    o2 = e.cd_run(seed=seed, first_index=512)
    assert e.last_cd_kernel() == 'cd_phase2_qs_kernel'
    assert np.array_equal(e.download(), Xq[:, 512:1024]) and np.array_equal(o2['f0'], oq['f0'][512:1024])
    # oracle trajectories
    prob = orc.Problem(funcs)
    for r in (0, 777, R - 1):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(r)
        x, s1, s2 = prob.improve_cd(X0[:, r], rng=rng)
        assert rel(Xq[:, r], x) < 1e-9, r
        assert oq['visits2'][r] == s2[1] and oq['accepted2'][r] == s2[2]
