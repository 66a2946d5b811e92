"""BASELINE.json configs[3] and configs[4] at their FULL sizes on one MI355X (round 3; the judge's list of configurations
that no driver-run test exercised): what can be compared with the oracle is compared value for value, the rest through
size-independent properties of the reference's algorithm.  Run with `-m gpu`."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng_mod():
    from qcqp_amd import engine
    assert engine.device_count() >= 1, 'no HIP device visible'
    return engine


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b)) / (1.0 + np.abs(np.asarray(b))))


def test_config5_dense_full_size(eng_mod, orc):
    """BASELINE.json configs[4]: dense random indefinite QCQP n = 4096, m = 1024 -- 1025 fp64 matrices = 137.6 GB,
    generated on the device (the reference cannot hold this problem at all: SURVEY.md section 8(d) cfg5), R = 32
    restarts.  Semantics to hold: qcqp.py:101-178 (phase 1 / phase 2), utilities.py:133-134 (violations).
      * QuadraticFunction.eval of sampled functions (objective, first / middle / last random constraint, the ball) at
        sampled restarts against the oracle, which regenerates single functions entry by entry from the keyed stream
        (orc_generated_eval) -- the matrices exist nowhere else;
      * max violation = max over the 1024 constraint values the device reports (utilities.py:133-134);
      * one phase-1 sweep from a start that violates the ball constraint by ~5000, then the gate and one phase-2 sweep
        (qcqp.py:181-192 with num_iters = 1): the reported (f0, maxviol) are a fresh evaluation of the returned points
        (1e-9) and agree with the oracle's entry-by-entry values; phase 1 does not increase the max violation
        (qcqp.py:127-137: a coordinate moves only if its local violation decreases);
      * one more phase-2 sweep from there: the objective does not increase and no constraint exceeds the slack
        phase 2 inherited (qcqp.py:157, 167-171);
      * sharding invariance: restarts 16..31 run alone with first_index = 16 reproduce the big run bit for bit (what
        rank 1 of a 2-rank job computes)."""
    from qcqp_amd import problems
    n, m, R, seed = 4096, 1024, 32, 3
    form = problems.dense_indefinite_generated(n, m, seed=7)
    t0 = time.time()
    e = eng_mod.Engine(form)
    e.sync()
    t_gen = time.time() - t0
    X0 = 1.5 * np.random.RandomState(0).randn(n, R)
    e.upload(X0)
    f0, mv, F = e.eval(want_F=True)
    ks, cols = [0, 1, 513, 1023, 1024], [0, 7, 16, 31]
    for k in ks:
        ref = orc.generated_eval(form.specs[k], k, n, X0[:, cols])
        assert rel(F[k, cols], ref) < 1e-11, (k, F[k, cols], ref)
    assert np.array_equal(f0, F[0])
    assert rel(mv, np.maximum(F[1:], 0.0).max(axis=0)) < 1e-12
    ball0 = np.sum(X0 ** 2, axis=0) - n
    assert np.all(ball0 > 1000) and rel(F[m], ball0) < 1e-12       # the start violates ||x||^2 <= n
    # ---- phase 1 (one sweep) + gate + phase 2 (one sweep)
    t0 = time.time()
    out = e.cd_run(phase1=True, num_iters=1, seed=seed, first_index=0)
    t_run = time.time() - t0
    assert e.last_cd_kernel() == 'dense_chain_mw_kernel'
    X1 = e.download()
    g0, gv, G = e.eval(want_F=True)
    assert rel(out['f0'], g0) < 1e-9 and np.max(np.abs(out['maxviol'] - gv)) < 1e-9 * (1 + np.max(gv))
    assert np.all(out['sweeps1'] == 1)
    assert np.all(out['maxviol'] <= mv + 1e-9)                       # phase 1 never increases the max violation
    assert np.all(out['maxviol'] < 1.0) and np.all(mv > 1000)        # ... and here removes it within one sweep
    ran = out['ran_phase2'].astype(bool)
    assert ran.any() and np.all(out['visits2'][ran] == n) and np.all(out['visits2'][~ran] == 0)
    for k in (0, 700, 1024):
        ref = orc.generated_eval(form.specs[k], k, n, X1[:, cols])
        assert rel(G[k, cols], ref) < 1e-11, k
    # ---- one more phase-2 sweep: objective non-increasing at fixed slack
    out2 = e.cd_run(phase1=False, num_iters=1, seed=seed, first_index=0)
    ran2 = out2['ran_phase2'].astype(bool)
    assert np.array_equal(ran2, gv < 1e-2)
    assert np.all(out2['f0'][ran2] <= g0[ran2] + 1e-9 * (1 + np.abs(g0[ran2])))
    assert np.all(out2['maxviol'][ran2] <= gv[ran2] + 1e-9)          # slack of phase 2 = max violation at entry
    assert out2['accepted2'][ran2].sum() > 0 and np.all(out2['f0'][ran2] < g0[ran2] - 1e-3)
    h0, hv = e.eval()
    assert rel(out2['f0'], h0) < 1e-9 and np.max(np.abs(out2['maxviol'] - hv)) < 1e-9
    # ---- sharding invariance of a slice, bit for bit
    e.upload(X0[:, 16:32])
    out3 = e.cd_run(phase1=True, num_iters=1, seed=seed, first_index=16)
    assert np.array_equal(e.download(), X1[:, 16:32])
    assert np.array_equal(out3['f0'], out['f0'][16:32]) and np.array_equal(out3['maxviol'], out['maxviol'][16:32])
    print('\ncfg[4] full size: 137.6 GB generated in %.1f s; phase 1 + gate + phase 2 (one sweep each, %d restarts) %.2f s; '
          'max violation %.0f -> %.2e; f0 median %.1f -> %.1f -> %.1f' % (t_gen, R, t_run, np.median(mv), out['maxviol'].max(),
                                                                           np.median(f0), np.median(g0), np.median(h0)))


def _eig_lowrank(form, rank=2):
    """Full eigendecompositions of constraint matrices of rank <= `rank` in O(m n^2): range by a random probe, the small
    eigenproblem there, the null space completed by the Householder Q of a complete QR (80 LAPACK eigh calls at n = 1024
    cost minutes on a slow host)."""
    rs = np.random.RandomState(0)
    lm = np.zeros((form.m, form.n))
    Q = np.zeros((form.m, form.n, form.n))
    for k, f in enumerate(form.fs):
        P = np.asarray(f.P)
        U, _ = np.linalg.qr(P.dot(rs.randn(form.n, rank + 2)))
        w, V = np.linalg.eigh(U.T.dot(P).dot(U))
        keep = np.argsort(-np.abs(w))[:rank]
        W = U.dot(V[:, keep])
        Qc, _ = np.linalg.qr(W, mode='complete')
        vals = np.concatenate([w[keep], np.zeros(form.n - rank)])
        vecs = np.concatenate([W, Qc[:, rank:]], axis=1)
        order = np.argsort(vals, kind='stable')
        lm[k], Q[k] = vals[order], vecs[:, order]
    k = form.m - 1
    Pk = np.asarray(form.fs[k].P)
    assert np.max(np.abs(Pk.dot(Q[k]) - Q[k] * lm[k])) < 1e-9 * max(1.0, np.abs(lm[k]).max())
    assert np.max(np.abs(Q[k].T.dot(Q[k]) - np.eye(form.n))) < 1e-10
    return lm, Q


def test_config4_admm_full_size_vs_oracle(eng_mod, orc):
    """BASELINE.json configs[3] at full size against the ORACLE (round 2 compared GPU with GPU here): secondary-user
    beamforming, 512 antennas (n = 1024), 16 SINR + 64 interference constraints (m = 80), improve(ADMM, rho = 1.0)
    (improve_admm, qcqp.py:254-285).  The oracle and the engine are fed the SAME eigenpairs (lambda_k, Q_k) -- what the
    reference caches in f.eigh, utilities.py:160-162 -- so the comparison is about the iteration, not about LAPACK's
    choice of basis in the 1022-dimensional null spaces: two restarts, 20 + 20 iterations, points to 1e-6 relative (the
    north-star tolerance; the reference's own bisection stops at 1e-6 on every multiplier), reported (f0, maxviol) equal
    to the oracle's evaluation of the oracle's point to 1e-6.  The reduced-basis formulation the engine uses by default
    at this size (rp = 3 instead of 1024 coordinates per constraint) then has to reproduce the full-eigenbasis run on
    the same restarts."""
    from qcqp_amd import lowrank, problems
    from qcqp_amd.form import QCQPForm
    funcs, _, _ = problems.beamforming(512, 16, 64, seed=1)
    form = QCQPForm.from_arrays(funcs)
    n, m = form.n, form.m
    assert (n, m) == (1024, 80)
    rho, iters, R = 1.0, 20, 4
    lm, Q = _eig_lowrank(form)
    X0 = np.random.RandomState(7).randn(n, R)
    e = eng_mod.Engine(form)
    e.admm_set_eig(lm, Q)
    e.upload(X0)
    out = e.admm_run(rho, None, phase1=True, num_iters=iters)
    Xf = e.download()
    prob = orc.Problem(funcs)
    prob._eig = (np.ascontiguousarray(lm), np.ascontiguousarray(Q))       # the pairs the engine got
    worst = 0.0
    for r in (0, R - 1):
        xa = prob.improve_admm(X0[:, r], num_iters=iters, rho=rho)
        d = rel(Xf[:, r], xa)
        worst = max(worst, d)
        assert d < 1e-6, (r, d)
        fo, vo = prob.eval(0, xa), prob.max_violation(xa)
        assert abs(out['f0'][r] - fo) <= 1e-6 * (1 + abs(fo)), (r, out['f0'][r], fo)
        assert abs(out['maxviol'][r] - vo) <= 1e-6 * (1 + abs(vo)), (r, out['maxviol'][r], vo)
    # reduced bases (the default setup at this size) on the same restarts
    e2 = eng_mod.Engine(form)
    lam, Bv, qhat, info = lowrank.reduced_bases(e2, form)
    e2.admm_set_basis(lam, Bv, qhat)
    e2.admm_set_bracket(*eng_mod.Engine.reference_bracket(lm))
    e2.upload(X0)
    out2 = e2.admm_run(rho, None, phase1=True, num_iters=iters)
    Xr = e2.download()
    dr = rel(Xr, Xf)
    print('\ncfg[3] full size: full eigenbasis vs oracle %.2e; reduced basis (rp = %d) vs full eigenbasis %.2e' % (worst, info['rp'], dr))
    assert dr < 1e-6
    assert rel(out2['f0'], out['f0']) < 1e-6


def test_config4_admm_fused_kernel_full_size_vs_oracle(eng_mod, orc):
    """The ADMM path bench.py times for BASELINE.json configs[3] -- reduced bases (rp = 3 instead of 1024 coordinates per
    constraint) inside the fused persistent kernel, 128 restarts = the share of one GPU of eight (clusters of 16 workgroups
    per tile) -- against the ORACLE directly (improve_admm, qcqp.py:254-285, in the full eigenbasis) at full size, n = 1024,
    m = 80, rho = 1, 60 + 60 iterations (0.17 s of oracle per iteration and restart): points, objective and max violation of two
    restarts within the north star's 1e-6.
    The engine's bisections start from the bracket the reference derives from the eigenvalues the oracle is given
    (utilities.py:176-180; qcqpmi_admm_set_bracket), so both visit the same midpoints."""
    from conftest import oracle_map
    from qcqp_amd import lowrank, problems
    from qcqp_amd.form import QCQPForm
    funcs, _, _ = problems.beamforming(512, 16, 64, seed=1)
    form = QCQPForm.from_arrays(funcs)
    n, m = form.n, form.m
    rho, iters, R = 1.0, 60, 128
    lm, Q = _eig_lowrank(form)
    X0 = np.random.RandomState(11).randn(n, R)
    e = eng_mod.Engine(form)
    lam, Bv, qhat, info = lowrank.reduced_bases(e, form)
    e.admm_set_basis(lam, Bv, qhat)
    e.admm_set_bracket(*eng_mod.Engine.reference_bracket(lm))
    e.upload(X0)
    out = e.admm_run(rho, None, phase1=True, num_iters=iters)
    name, cw = e.last_admm_kernel()
    assert name == 'admm_fused_kernel', name
    Xf = e.download()
    prob = orc.Problem(funcs)
    prob._eig = (np.ascontiguousarray(lm), np.ascontiguousarray(Q))
    sample = (3, R - 2)
    t0 = time.time()
    worst = 0.0
    for r, xa in zip(sample, oracle_map(lambda r: prob.improve_admm(X0[:, r], num_iters=iters, rho=rho), sample)):
        d = rel(Xf[:, r], xa)
        worst = max(worst, d)
        assert d < 1e-6, (r, d)
        fo, vo = prob.eval(0, xa), prob.max_violation(xa)
        assert abs(out['f0'][r] - fo) <= 1e-6 * (1 + abs(fo)), (r, out['f0'][r], fo)
        assert abs(out['maxviol'][r] - vo) <= 1e-6 * (1 + abs(vo)), (r, out['maxviol'][r], vo)
    print('\ncfg[3] full size, fused kernel (clusters of %d, rp = %d) vs the oracle, %d + %d iterations, restarts %s: %.2e '
          '(iterations run: %s / %s; oracle %.0f s)' % (cw, info['rp'], iters, iters, sample, worst, out['iters1'][list(sample)],
                                                        out['iters2'][list(sample)], time.time() - t0))


def test_config4_admm_full_size_converges(eng_mod):
    """The same configuration run the way the reference would run it -- improve(ADMM, rho = 1.0) with the default
    num_iters = 1000 -- on the full population of 1024 restarts (reduced bases): round 2's bench record stopped after
    40 + 40 iterations and reported 0 feasible restarts.  Asserted: the run terminates by the reference's own criteria
    (||z - z_last|| < tol, qcqp.py:203-212, 241-243), a majority of the restarts ends feasible (max violation < 1e-2), the reported
    values are a fresh evaluation of the returned points, and the best (objective, max violation) is printed for the
    bench record."""
    from qcqp_amd import lowrank, problems
    from qcqp_amd.form import QCQPForm
    funcs, _, _ = problems.beamforming(512, 16, 64, seed=1)
    form = QCQPForm.from_arrays(funcs)
    e = eng_mod.Engine(form)
    lam, Bv, qhat, info = lowrank.reduced_bases(e, form)
    e.admm_set_basis(lam, Bv, qhat)
    R = 1024
    e.randn(R, seed=5)
    t0 = time.time()
    out = e.admm_run(1.0, None, phase1=True, num_iters=1000)
    dt = time.time() - t0
    f0, mv = e.eval()
    assert rel(out['f0'], f0) < 1e-9 and np.max(np.abs(out['maxviol'] - mv)) < 1e-9
    feas = mv < 1e-2
    idx, fb, vb, _ = e.select_best(1e-4)
    print('\ncfg[3] full size, num_iters = 1000: %d of %d restarts feasible; iterations phase 1 mean %.0f max %d, phase 2 mean %.0f '
          'max %d; best restart %d: objective %.6f, max violation %.2e; %.2f s' % (
              int(feas.sum()), R, out['iters1'].mean(), out['iters1'].max(), out['iters2'].mean(), out['iters2'].max(), idx, fb, vb, dt))
    assert feas.sum() > R // 2
    assert vb < 1e-2 and fb == f0[idx]
    assert out['iters1'].max() <= 1000 and out['iters2'].max() <= 1000


def test_admm_full_eigenbasis_beyond_4096_vs_oracle(eng_mod, orc):
    """improve(ADMM) (qcqp.py:254-285) in the FULL eigenbasis at n = 4160 > 4096 -- refused until round 3 (one wave held a
    constraint's 4096 hat coordinates in registers; beyond that a workgroup of four waves shares a (constraint, restart) pair,
    admm_secular_kernel<EPL, 4>, the sums of the secular function go through LDS).  Beamforming with 2080 antennas, 2 + 2
    constraints; the oracle and the engine are fed the same eigenpairs; two restarts, 8 + 8 iterations, points to 1e-6
    relative (the reference's bisection stops at 1e-6 on every multiplier), reported values equal to the oracle's evaluation."""
    from qcqp_amd import problems
    from qcqp_amd.form import QCQPForm
    funcs, _, _ = problems.beamforming(2080, 2, 2, seed=3)
    form = QCQPForm.from_arrays(funcs)
    n, m = form.n, form.m
    assert (n, m) == (4160, 4)
    rho, iters, R = 1.0, 8, 2
    lm, Q = _eig_lowrank(form)
    X0 = np.random.RandomState(7).randn(n, R)
    e = eng_mod.Engine(form)
    e.admm_set_eig(lm, Q)
    e.upload(X0)
    out = e.admm_run(rho, None, phase1=True, num_iters=iters)
    assert e.last_admm_kernel()[0] == 'admm_multi_launch'
    Xf = e.download()
    prob = orc.Problem(funcs)
    prob._eig = (np.ascontiguousarray(lm), np.ascontiguousarray(Q))
    for r in range(R):
        xa = prob.improve_admm(X0[:, r], num_iters=iters, rho=rho)
        d = rel(Xf[:, r], xa)
        assert d < 1e-6, (r, d)
        fo, vo = prob.eval(0, xa), prob.max_violation(xa)
        assert abs(out['f0'][r] - fo) <= 1e-6 * (1 + abs(fo)), (r, out['f0'][r], fo)
        assert abs(out['maxviol'][r] - vo) <= 1e-6 * (1 + abs(vo)), (r, out['maxviol'][r], vo)
