"""GPU parity tests at the sizes where the dispatch changes: the LDS-tiled MFMA sampler / evaluator
(BASELINE.json configs[2] at full size), the dense-constraint path with and without K-split, the unit
operators of the C ABI against the reference's golden vectors (G3 / G4), the rocSOLVER setup path.
Run with `-m gpu` on an MI355X."""
import ctypes as C

import numpy as np
import pytest

from conftest import RELSTR, load_golden, oracle_map

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



# ------------------------------------------------------------------ SDR sampling: the GEMM path
@pytest.mark.parametrize('n,S', [(200, 256), (500, 384), (2000, 256)])
def test_sdr_sample_gemm_path(eng_mod, orc, n, S):
    """x = mu + F xi through dense_products_kernel<3> (taken when n >= 113 and S >= 128; the n = 10 test of
    test_gpu_parity.py goes through affine_tiles_kernel): given normals and the device's keyed Philox
    normals, against the plain NumPy product (qcqp.py:396 with the factor hoisted)."""
    from qcqp_amd import problems
    funcs, _, _ = problems.maxcut(n, 0.5, seed=4)
    e = make(eng_mod, funcs)
    rs = np.random.RandomState(n)
    F = rs.randn(n, n) / np.sqrt(n)
    mu = rs.randn(n)
    Xi = rs.randn(n, S)
    e.sdr_sample(mu, F, S, Xi=Xi)
    X = e.download()
    ref = mu[:, None] + F.dot(Xi)
    assert np.max(np.abs(X - ref)) < 1e-12 * (1 + np.max(np.abs(ref)))
    e.sdr_sample(mu, F, S, seed=11, first_index=77)
    Xd = e.download()
    Xi2 = orc.keyed_normal_matrix(11, n, S, first_index=77)
    ref2 = mu[:, None] + F.dot(Xi2)
    assert np.max(np.abs(Xd - ref2)) < 1e-11 * (1 + np.max(np.abs(ref2)))


def test_config3_maxcut_full_size(eng_mod, orc):
    """BASELINE.json configs[2] at full size: MAXCUT on G(2000, 0.5), 8192 Goemans-Williamson samples
    x ~ N(0, X*) with X* = V V^T a unit-diagonal PSD matrix (rank 40, the shape an SDP solution has),
    sampled and evaluated on the device.  Objective and max violation of 40 columns against the oracle
    (1e-12), sample covariance against X*, sharding invariance (samples 1024..2047 drawn alone with their
    global index offset are bit-identical: what rank 1 of 8 computes)."""
    from qcqp_amd import problems
    n, S, rk = 2000, 8192, 40
    funcs, _, _ = problems.maxcut(n, 0.5, seed=1)
    e = make(eng_mod, funcs)
    rs = np.random.RandomState(5)
    V = rs.randn(n, rk)
    V /= np.linalg.norm(V, axis=1)[:, None]
    F = np.zeros((n, n))
    F[:, :rk] = V                      # x = F xi has covariance V V^T
    mu = np.zeros(n)
    e.sdr_sample(mu, F, S, seed=2024, first_index=0)
    f0, mv = e.eval()
    X = e.download()
    # the sampler against the oracle's keyed normals on a slice of columns
    cols = np.r_[0:16, 1024:1040, 8184:8192]
    Xi = np.stack([orc.keyed_normal_matrix(2024, n, 1, first_index=int(s))[:, 0] for s in cols], axis=1)
    assert np.max(np.abs(X[:, cols] - V.dot(Xi[:rk]))) < 1e-11
    # evaluation (qcqp.py:399-401) of those columns against the oracle
    prob = orc.Problem(funcs)
    g0, gv = prob.eval_batch(X[:, cols])
    assert rel(f0[cols], g0) < 1e-12 and rel(mv[cols], gv) < 1e-12
    # distribution: sample covariance of 8192 draws vs X* (standard error ~ 1/sqrt(S) = 0.011 per entry)
    sub = np.arange(0, n, 40)
    Cs = X[sub].dot(X[sub].T) / S
    assert np.max(np.abs(Cs - V[sub].dot(V[sub].T))) < 0.08
    assert abs(np.mean(X)) < 5e-3
    # sharding invariance: the share of rank 1 of 8
    e.sdr_sample(mu, F, 1024, seed=2024, first_index=1024)
    Xs = e.download()
    assert np.array_equal(Xs, X[:, 1024:2048])
    f1, m1 = e.eval()
    assert np.array_equal(f1, f0[1024:2048]) and np.array_equal(m1, mv[1024:2048])
    # the one-call entry (qcqpmi_sdr_sample_eval): 8192 samples in chunks of 2048 through two reused buffers, no
    # population laid out -- the same values bit for bit; the winner re-drawn from its index is the same point
    f2, m2 = e.sdr_sample_eval(mu, F, S, seed=2024, first_index=0)
    assert np.array_equal(f2, f0) and np.array_equal(m2, mv)
    assert e.pop_size == 2048
    w = int(np.argmin(f0))
    e.sdr_sample(None, None, 1, seed=2024, first_index=w)
    assert np.array_equal(e.download()[:, 0], X[:, w])


@pytest.mark.parametrize('n,S,chunk', [(200, 1000, 256), (10, 77, 16), (500, 384, 0)])
def test_sdr_sample_eval_one_call(eng_mod, orc, n, S, chunk, monkeypatch):
    """qcqpmi_sdr_sample_eval (qcqp.py:396-401 for S samples in one call): several chunks (a ragged last one), with and without
    the points, against the oracle's keyed normals + NumPy's product + the oracle's evaluation, and bit-identical to
    qcqpmi_pop_sdr_sample + qcqpmi_pop_eval on the whole population."""
    from qcqp_amd import problems
    funcs, _, _ = problems.boolean_least_squares(n, max(4, n // 4), seed=6)
    e = make(eng_mod, funcs)
    rs = np.random.RandomState(n + S)
    F = rs.randn(n, n) / np.sqrt(n)
    mu = rs.randn(n)
    e.sdr_sample(mu, F, S, seed=5, first_index=300)
    fa, ma = e.eval()
    Xa = e.download()
    if chunk:
        monkeypatch.setenv('QCQPMI_SDR_CHUNK', str(chunk))
    f0, mv, X = e.sdr_sample_eval(mu, F, S, seed=5, first_index=300, want_X=True)
    assert np.array_equal(X, Xa) and np.array_equal(f0, fa) and np.array_equal(mv, ma)
    f1, m1 = e.sdr_sample_eval(None, None, S, seed=5, first_index=300)          # the resident pair, the points not kept
    assert np.array_equal(f1, fa) and np.array_equal(m1, ma)
    ref = mu[:, None] + F.dot(orc.keyed_normal_matrix(5, n, S, first_index=300))
    assert np.max(np.abs(X - ref)) < 1e-11 * (1 + np.max(np.abs(ref)))
    g0, gv = orc.Problem(funcs).eval_batch(X)
    assert rel(f0, g0) < 1e-12 and rel(mv, gv) < 1e-12


# ------------------------------------------------------------------ dense path at dispatch-changing sizes
@pytest.mark.parametrize('R', [48, 4096])
def test_dense_path_scale_dispatch_vs_oracle(eng_mod, orc, R):
    """BASELINE.json configs[4] family (dense indefinite constraints) at n = 256, m = 130: R = 48 takes the
    K-split products (grid.z = 7 partial planes + the fix-up plane) with G staged in LDS; R = 4096 takes one
    plane and the occupancy-bound chain.  Both overlap the chain of block b with the products of block b + 1 on
    a second stream (hole + fix-up launch).

    What can be asserted: phase 1 drives every coordinate to a point where the feasible set of the bisection is
    nearly degenerate, so a 1e-13 difference in a tracked f_k (MFMA summation order vs the oracle's sequential
    sums) is amplified by 1/sqrt(discriminant) per coordinate: the oracle itself (tests/test_host_cpu.py::test_reference_cd_is_chaotic_under_one_ulp) shows a quarter of the restarts
    1e-9 off the oracle after ONE phase-1 sweep and drifting apart from there, for every R and K-split alike.
    Per-restart trajectories are therefore compared as outcomes, not bit patterns: every restart's reported
    values are a fresh evaluation of its point (exact), sampled restarts end in the same feasibility class as
    the oracle's and with an objective from the same distribution; the number that stayed on the oracle's
    trajectory is printed."""
    from qcqp_amd import problems
    n, m, iters, seed, first = 256, 130, 2, 13, 5
    funcs, _, _ = problems.dense_indefinite(n, m, seed=11)
    e = make(eng_mod, funcs)
    prob = orc.Problem(funcs)
    X0 = 1.5 * np.random.RandomState(3).randn(n, R)
    e.upload(X0)
    out = e.cd_run(phase1=True, num_iters=iters, seed=seed, first_index=first)
    X = e.download()
    f0, mv = e.eval()
    assert rel(out['f0'], f0) < 1e-9 and np.max(np.abs(out['maxviol'] - mv)) < 1e-9
    sample = [0, R // 2, R - 1]
    same = 0
    spread = np.percentile(f0, 90) - np.percentile(f0, 10)

    def oracle_restart(r):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(first + r)
        return prob.improve_cd(X0[:, r], num_iters=iters, rng=rng)
    trajectories = dict(zip(sample, oracle_map(oracle_restart, sample)))      # 25 s of oracle each: side by side, and once
    for r in sample:
        x, s1, s2 = trajectories[r]
        d = np.max(np.abs(X[:, r] - x))
        same += d < 1e-6 * (1 + np.max(np.abs(x)))
        fo, mo = prob.eval(0, x), prob.max_violation(x)
        # the device's own report about its point is exact
        assert abs(prob.eval(0, X[:, r]) - out['f0'][r]) <= 1e-9 * (1 + abs(out['f0'][r])), r
        assert abs(prob.max_violation(X[:, r]) - out['maxviol'][r]) < 1e-9, r
        # same outcome class as the oracle's restart
        assert (out['maxviol'][r] < 1e-2) == (mo < 1e-2), (r, out['maxviol'][r], mo)
        assert abs(out['f0'][r] - fo) <= max(spread, 0.05 * abs(fo)), (r, out['f0'][r], fo, spread)
        assert abs(int(out['sweeps1'][r]) - int(s1[0])) <= 1, r
    print('\ndense n=256 m=130 R=%d: %d of %d sampled restarts on the oracle trajectory (1e-6)' % (R, same, len(sample)))
    if R == 48:
        # the reference-order counterpart (qcqpmi_cd_reference_order): the same restarts VALUE FOR VALUE -- with ONE sweep per phase
        # (the diagnostic kernel needs 16 s per sweep of this problem whatever the number of restarts: 64 s of the suite with two)
        def oracle_restart1(r):
            rng = orc.Rng(orc.RNG_KEYED, seed)
            rng.set_restart(first + r)
            return prob.improve_cd(X0[:, r], num_iters=1, rng=rng)
        trajectories1 = dict(zip(sample, oracle_map(oracle_restart1, sample)))
        e.cd_reference_order(True)
        e.upload(X0)
        outr = e.cd_run(phase1=True, num_iters=1, seed=seed, first_index=first)
        assert e.last_cd_kernel() == 'cd_general_kernel'
        Xr = e.download()
        for r in sample:
            x, s1, s2 = trajectories1[r]
            assert rel(Xr[:, r], x) < 1e-9, (r, np.max(np.abs(Xr[:, r] - x)))
            assert outr['sweeps1'][r] == s1[0] and outr['visits2'][r] == s2[1] and outr['accepted2'][r] == s2[2], r
            assert abs(outr['f0'][r] - prob.eval(0, x)) <= 1e-9 * (1 + abs(outr['f0'][r]))


@pytest.mark.parametrize('family', ['dense100', 'dense128', 'beam100'])
def test_population_best_vs_oracle(eng_mod, orc, family):
    """The north-star statement on the dense-constraint path: same inputs, same keyed
    streams, R = 512 restarts through the GPU and through the oracle; compare the best (objective, max
    violation) of the two populations (qcqp.py:252-254 ordering) and print the per-restart divergence rate.

    Round 3: the population ALSO runs through the reference-order mode of the engine (qcqpmi_cd_reference_order: the
    one-variable coefficients summed like the reference sums them), which has to follow the oracle restart by restart to
    1e-9 -- that is the value-level parity statement for coupled constraints at n > 64, well inside the north star's
    1e-6.  What follows about the fast MFMA path is judged against that counterpart as a distribution.

    (Round 4: WHY the fast path cannot be held to the oracle's free-running trajectories -- the reference itself moves 41 % / 100 %
    of its restarts by more than 1e-6 under one ulp of input, profiles/r04_reference_sensitivity.md -- and the value-level
    statement that replaces it: test_dense_default_path_follows_the_oracle_step_by_step, every visit within 1e-9 of the oracle
    on the oracle's own states.)

    dense_indefinite (phase 2 dominates): restarts that leave the oracle trajectory stay in its basin, the best
    restart is the same one and its objective agrees to 1e-5 relative (measured 1.4e-6 / 3.8e-8: a restart that forks
    ends within the move tolerance tol = 1e-4 of the same local minimiser, an O(tol^2)-O(tol) difference in f).
    beamforming (rank-2 constraints, phase 1 dominates): the bisection end points are degenerate and every restart
    is chaotic after a few coordinates (see the test above); the populations agree as distributions (feasible
    count, quartiles, best objective within the quartile noise), which is all the reference's own rerun with a
    different BLAS would give."""
    from qcqp_amd import problems
    R, iters, seed, first = 512, 5, 13, 5
    if family == 'dense100':
        funcs = problems.dense_indefinite(100, 30, seed=11)[0]
    elif family == 'dense128':
        funcs = problems.dense_indefinite(128, 40, seed=12)[0]
        R = 256      # 0.19 s of oracle per restart
    else:
        funcs = problems.beamforming(50, 12, 4, seed=3)[0]
    n = funcs[0][0].shape[0]
    e = make(eng_mod, funcs)
    prob = orc.Problem(funcs)
    X0 = 1.5 * np.random.RandomState(3).randn(n, R)
    e.upload(X0)
    out = e.cd_run(phase1=True, num_iters=iters, seed=seed, first_index=first)
    X = e.download()
    f0, mv = e.eval()
    assert rel(out['f0'], f0) < 1e-9 and np.max(np.abs(out['maxviol'] - mv)) < 1e-9
    Xo, fo, mo = np.zeros_like(X), np.zeros(R), np.zeros(R)

    def oracle_restart(r, x0=None):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(first + r)
        x = prob.improve_cd(X0[:, r] if x0 is None else x0, num_iters=iters, rng=rng)[0]
        return x, prob.eval(0, x), prob.max_violation(x)
    for r, (x, f_, m_) in enumerate(oracle_map(oracle_restart, range(R))):      # (the C oracle runs without the GIL)
        Xo[:, r], fo[r], mo[r] = x, f_, m_

    def best(f, v):
        feas = np.where(v < 1e-2)[0]
        i = feas[np.argmin(f[feas])] if len(feas) else int(np.argmin(v))
        return int(i), f[i], v[i]

    d = np.max(np.abs(X - Xo), axis=0) / (1 + np.max(np.abs(Xo), axis=0))
    off = float(np.mean(d > 1e-6))
    ig, fg, vg = best(f0, mv)
    io, fb, vb = best(fo, mo)
    relbest = abs(fg - fb) / (1 + abs(fb))
    print('\n%s R=%d: %.1f %% of restarts leave the oracle trajectory (1e-6); best GPU restart %d f0 %.10g maxviol %.2e, '
          'best oracle restart %d f0 %.10g maxviol %.2e, relative difference %.2e' % (family, R, 100 * off, ig, fg, vg, io, fb, vb, relbest))
    assert (vg < 1e-2) == (vb < 1e-2)
    assert abs(int((mv < 1e-2).sum()) - int((mo < 1e-2).sum())) <= max(2, R // 50)
    # ---- value-level parity: the same population through the reference-order mode (row-sequential sums like the
    # reference's CSR products, t0 = f_k(z) afresh per coordinate) must FOLLOW THE ORACLE restart by restart -- chaotic
    # families included -- and therefore pick the same best restart with the same (objective, max violation)
    e.cd_reference_order(True)
    e.upload(X0)
    outr = e.cd_run(phase1=True, num_iters=iters, seed=seed, first_index=first)
    assert e.last_cd_kernel() == 'cd_general_kernel'
    Xr = e.download()
    e.cd_reference_order(False)
    dr = np.max(np.abs(Xr - Xo), axis=0) / (1 + np.max(np.abs(Xo), axis=0))
    ir, fr_, vr = best(outr['f0'], outr['maxviol'])
    print('%s reference-order mode: worst restart %.2e off the oracle; best restart %d f0 %.12g (oracle %d, %.12g)'
          % (family, dr.max(), ir, fr_, io, fb))
    assert dr.max() < 1e-9, (int(np.argmax(dr)), dr.max())
    assert ir == io and abs(fr_ - fb) <= 1e-9 * (1 + abs(fb)) and abs(vr - vb) <= 1e-9
    assert rel(outr['f0'], fo) < 1e-9 and np.max(np.abs(outr['maxviol'] - mo)) < 1e-9
    # ---- the fast (MFMA) path against that counterpart: a distribution statement (see the docstring)
    qg, qo = np.percentile(f0, [25, 50, 75]), np.percentile(fo, [25, 50, 75])
    assert np.all(np.abs(qg - qo) <= 0.05 * (qo[2] - qo[0]) + 0.02 * np.abs(qo)), (qg, qo)
    # the yardstick for `off`: how many of the ORACLE's restarts leave the oracle's own trajectory when x0 moves by one ulp
    # (the reference's arithmetic under the smallest possible input change; see tests/test_host_cpu.py and
    # profiles/r04_reference_sensitivity.md for the same experiment with /root/reference)
    Rs = min(R, 96)
    dref = np.zeros(Rs)
    for r, (xp, _, _) in enumerate(oracle_map(lambda r: oracle_restart(r, np.nextafter(X0[:, r], np.inf)), range(Rs))):
        dref[r] = np.max(np.abs(xp - Xo[:, r])) / (1 + np.max(np.abs(Xo[:, r])))
    off_ref = float(np.mean(dref > 1e-6))
    print('%s: the oracle against itself under one ulp of x0: %.1f %% of %d restarts leave the trajectory (1e-6); the MFMA path against the '
          'oracle: %.1f %%' % (family, 100 * off_ref, Rs, 100 * off))
    assert off <= off_ref + 0.25, (off, off_ref)      # the fast path leaves the oracle no more often than the oracle leaves itself
    if family.startswith('dense'):
        assert off < 0.75, off
        assert ig == io, (ig, io)
        assert relbest < 1e-5, relbest
    else:
        assert relbest < 0.10, relbest      # best of 512 chaotic draws: order-statistic noise, measured 3.9e-2


@pytest.mark.parametrize('family', ['dense100', 'dense128', 'beam100'])
def test_dense_default_path_follows_the_oracle_step_by_step(eng_mod, orc, family):
    """VALUE-LEVEL parity of the DEFAULT coupled-constraint path (MFMA products + dense_chain_mw_kernel) with the reference,
    in the only form the reference's own dynamics allow.  Coordinate descent with coupled constraints is chaotic in the
    reference itself: moving x0 by ONE ULP sends 41 % (dense family) / 100 % (beamforming) of the reference's restarts more
    than 1e-6 away (profiles/r04_reference_sensitivity.md, measured with /root/reference; the same experiment through the
    oracle is tests/test_host_cpu.py::test_reference_cd_is_chaotic_under_one_ulp), so a free-running trajectory of ANY
    arithmetic that is not bit-identical to SciPy's row-sequential sums leaves the reference's -- that comparison is made
    with the engine's reference-order mode (test_population_best_vs_oracle, 1e-9, every restart).  The fast path is
    compared here the way one compares maps of a chaotic system: TEACHER-FORCED.  The oracle records every state it visits
    (oracle.improve_cd_traced); the engine is handed the oracle's states of all restarts and runs its unit step
    (qcqpmi_cd_dense_block_step: fresh function values, products of the block on the matrix cores, the chain kernel with the
    keyed draws of that sweep, the phase-2 slack of the oracle's run):

    * VISIT BY VISIT -- one coordinate per step, every visit of every restart of phase 1 and phase 2: the new x_i must be the
      oracle's to 1e-6 relative (the north-star tolerance), bisection decisions (qcqp.py:122-131), accept tests (qcqp.py:132,
      168), interval rules and draws included -- asserted for EVERY visit;
    * BLOCK BY BLOCK -- the kernel's own unit, 16 visits per launch with the in-block Gauss-Seidel corrections: the first
      visit sees the oracle's state, the other 15 the engine's own moves, so the reference's sensitivity already acts inside
      the block (beamforming: a root moves by 1 / sqrt(discriminant) of the input noise); asserted as a share, printed."""
    from qcqp_amd import problems
    R, iters, seed, first = 64, 5, 13, 5
    if family == 'dense100':
        funcs = problems.dense_indefinite(100, 30, seed=11)[0]
    elif family == 'dense128':
        funcs = problems.dense_indefinite(128, 40, seed=12)[0]
        R = 32
    else:
        funcs = problems.beamforming(50, 12, 4, seed=3)[0]
    n = funcs[0][0].shape[0]
    NB = (n + 15) // 16
    e = make(eng_mod, funcs)
    prob = orc.Problem(funcs)
    X0 = 1.5 * np.random.RandomState(3).randn(n, R)
    runs = []
    for r in range(R):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(first + r)
        runs.append(prob.improve_cd_traced(X0[:, r], num_iters=iters, rng=rng))
    slack2 = np.array([0.0 if u[5] is None else u[5] for u in runs])

    def walk(width, far=None):
        """Teacher-forced walk along the oracle's trajectories in steps of `width` coordinates (1 or 16); `far` collects the
        steps whose result is more than 1e-6 away from the oracle's, with the state they started from."""
        cur = X0.copy()
        ds, steps, worst = [], 0, (0.0, None)
        for phase, ti in ((1, 3), (2, 4)):
            trs = [u[ti] for u in runs]
            nsweeps = (max(len(t) for t in trs) + n - 1) // n
            for t in range(nsweeps):
                for b in range(NB):
                    for c in range(0, min(16, n - 16 * b), width):
                        v0, nc = t * n + 16 * b + c, min(width, n - 16 * b - c)
                        active = [r for r in range(R) if len(trs[r]) > v0]
                        if not active:
                            continue
                        e.upload(cur)
                        e.cd_dense_block_step(phase, t, b, slack=slack2 if phase == 2 else None, seed=seed, first_index=first,
                                              coords=(c, c + width))
                        X1 = e.download()
                        steps += 1
                        i0 = 16 * b + c
                        for r in active:
                            nv = min(len(trs[r]) - v0, nc)
                            exp = trs[r][v0:v0 + nv]
                            d = np.max(np.abs(X1[i0:i0 + nv, r] - exp)) / (1 + np.max(np.abs(cur[:, r])))
                            ds.append(d)
                            if d > worst[0]:
                                worst = (d, (phase, t, b, c, r))
                            if far is not None and d > 1e-6:
                                far.append((d, phase, t, b, r, cur[:, r].copy()))
                            cur[i0:i0 + nv, r] = exp
        for r in range(R):
            assert np.array_equal(cur[:, r], runs[r][0]), r        # the states that were fed are the oracle's trajectory
        return np.array(ds), steps, worst

    ds, steps, worst = walk(1)
    assert e.last_cd_kernel() == 'dense_chain_mw_kernel'
    print('\n%s visit by visit: %d unit steps, %d visits compared with the oracle\'s: median %.1e, 99.9 %% %.1e, max %.1e at '
          '(phase, sweep, block, coordinate, restart) = %s; beyond 1e-6: %d, beyond 1e-9: %d' % (
              family, steps, len(ds), np.median(ds), np.percentile(ds, 99.9), ds.max(), worst[1], int((ds > 1e-6).sum()), int((ds > 1e-9).sum())))
    assert len(ds) >= 2 * n * R        # at least a phase-1 and a phase-2 sweep of every restart
    assert int((ds > 1e-6).sum()) == 0, worst            # the north-star tolerance, every visit
    assert int((ds > 1e-8).sum()) == 0, worst            # (measured: max 1.2e-10)
    assert np.median(ds) < 1e-12
    far = []
    db, steps, worst = walk(16, far)
    print('%s block by block: %d unit steps, %d blocks compared: median %.1e, 99 %% %.1e, max %.1e at %s; beyond 1e-6: %d (%.2f %%), beyond 1e-9: %d' % (
        family, steps, len(db), np.median(db), np.percentile(db, 99), db.max(), worst[1], int((db > 1e-6).sum()), 100 * np.mean(db > 1e-6), int((db > 1e-9).sum())))
    assert np.median(db) < 1e-11
    assert np.mean(db > 1e-6) < (0.005 if family.startswith('dense') else 0.05), np.mean(db > 1e-6)
    # THE YARDSTICK for every block beyond 1e-6: the oracle against ITSELF on that block from start states 1, 32 and 1024 ulps away
    # (every coordinate scaled by 1 +- k 2^-52; same draws, same slack).  The engine's coefficients differ from the oracle's by the
    # rounding of another summation order and of TRACKED function values -- up to ~1e-13 relative, i.e. hundreds of ulps --, so a
    # block where the reference's own map amplifies rounding (beamforming: 1 ulp in, 1e-5 out: a factor 1e11) is a block where
    # the oracle leaves itself by as much as the engine leaves it; a bug in the in-block Gauss-Seidel correction would show as an
    # engine deviation far beyond anything the oracle does to itself.  Asserted on every such block: engine <= 10 x the largest
    # of the oracle's own deviations (+ 1e-6); the three levels are printed.
    ratios, levels = [], {1: [], 32: [], 1024: []}
    for d, phase, t, b, r, start in far:
        i0, cnt = 16 * b, min(16, n - 16 * b)

        def block_from(st):
            rng = orc.Rng(orc.RNG_KEYED, seed)
            rng.set_restart(first + r)
            return prob.cd_visits(phase, st, t, i0, cnt, slack2=slack2[r], rng=rng)[i0:i0 + cnt]
        base = block_from(start)
        own = 0.0
        for k in (1, 32, 1024):
            dk = max(np.max(np.abs(block_from(start * (1.0 + sg * k * 2.0 ** -52)) - base)) for sg in (1.0, -1.0)) / (1 + np.max(np.abs(start)))
            levels[k].append(dk)
            own = max(own, dk)
        ratios.append(d / max(own, 1e-300))
        assert d <= 10.0 * own + 1e-6, (family, phase, t, b, r, d, own)
    if ratios:
        print('%s: %d blocks beyond 1e-6; the oracle against itself on them under 1 / 32 / 1024 ulps: median %.1e / %.1e / %.1e; '
              'engine deviation / largest oracle self-deviation: median %.2f, max %.2f'
              % (family, len(ratios), np.median(levels[1]), np.median(levels[32]), np.median(levels[1024]), np.median(ratios), max(ratios)))


def test_dense_default_path_follows_the_oracle_at_1024_by_256(eng_mod, orc):
    """The same teacher-forced comparison at the size of BASELINE.json configs[4]'s family that the dense path's secondary
    record runs -- n = 1024, m = 256: K-split products, dense_chain_mw_kernel with up to eight waves per restart, a launch
    geometry the n <= 128 cases never reach.  A sweep of the oracle costs two minutes per restart at this size (257 dense
    1024 x 1024 functions: get_onevar_func forms P_k z for every function at every visit, utilities.py:99-105), so the walk
    follows the FIRST TWO BLOCKS (32 visits: 0.6 s of oracle each) of a phase-1 sweep (from random starts) and of a phase-2 sweep
    (from feasible starts, slack 0) of 2 restarts, visit by visit: every new x_i within 1e-8 of the oracle's; then block by block (the
    kernel's own unit)."""
    from qcqp_amd import problems
    n, m, R, seed, first, NVIS = 1024, 256, 2, 17, 9, 32
    funcs = problems.dense_indefinite(n, m, seed=21)[0]
    e = make(eng_mod, funcs)
    prob = orc.Problem(funcs)
    rs = np.random.RandomState(8)
    X1 = 1.5 * rs.randn(n, R)                      # phase 1: infeasible random starts
    X2 = rs.randn(n, R)
    for r in range(R):                             # phase 2: feasible starts, as far out as the constraints allow
        a = 4.0
        while prob.max_violation(a * X2[:, r]) > 0.0:
            a *= 0.7
        X2[:, r] *= a
    for phase, X0 in ((1, X1), (2, X2)):
        trs, slack = [], np.zeros(R)
        for r in range(R):
            rng = orc.Rng(orc.RNG_KEYED, seed)
            rng.set_restart(first + r)
            _, tr, sl = prob.cd_phase_traced(phase, X0[:, r], NVIS, rng=rng)
            assert len(tr) == NVIS
            trs.append(tr)
            slack[r] = 0.0 if sl is None else sl
        assert phase == 1 or np.all(slack == 0.0)
        for width in (1, 16):
            cur = X0.copy()
            worst = 0.0
            for b in range(NVIS // 16):
                for c in range(0, 16, width):
                    e.upload(cur)
                    e.cd_dense_block_step(phase, 0, b, slack=slack if phase == 2 else None, seed=seed, first_index=first, coords=(c, c + width))
                    Xn = e.download()
                    i0 = 16 * b + c
                    for r in range(R):
                        exp = trs[r][i0:i0 + width]
                        d = np.max(np.abs(Xn[i0:i0 + width, r] - exp)) / (1 + np.max(np.abs(cur[:, r])))
                        worst = max(worst, d)
                        cur[i0:i0 + width, r] = exp
            assert e.last_cd_kernel() == 'dense_chain_mw_kernel'
            moved = sum(int(np.sum(np.abs(trs[r] - X0[:NVIS, r]) > 0)) for r in range(R))
            print('n = 1024, m = 256, phase %d, steps of %d: %d visits of %d restarts (%d of them moved), max deviation from the oracle %.1e'
                  % (phase, width, NVIS, R, moved, worst))
            assert moved >= NVIS // 4       # the walk is not a walk over fixed points
            assert worst < (1e-8 if width == 1 else 1e-6), (phase, width, worst)


# ------------------------------------------------------------------ unit operators vs the reference's goldens
def test_g3_feasible_intervals_on_device(eng_mod):
    """get_feasible_intervals (utilities.py:198-232) evaluated by the device function of onevar.h on the 4 800
    golden cases captured from the reference: bit-exact end points (same IEEE operations in the same order)."""
    from qcqp_amd import _ffi
    z = load_golden('g3_intervals')
    cases, out = z['cases'], z['out']
    N = len(cases)
    pqrs = np.ascontiguousarray(cases[:, :4])
    relop = np.ascontiguousarray(cases[:, 4].astype(np.int32))
    res = np.zeros((N, 5))
    rc = _ffi.lib().qcqpmi_feasible_intervals_batch(0, N, pqrs.ctypes.data_as(_ffi.c_dp),
                                                     relop.ctypes.data_as(C.POINTER(C.c_int)), res.ctypes.data_as(_ffi.c_dp))
    assert rc == 0
    for i in range(N):
        cnt = int(out[i, 0])
        assert int(res[i, 0]) == cnt, (i, cases[i])
        for j in range(cnt):
            assert res[i, 1 + 2 * j] == out[i, 1 + 2 * j] and res[i, 2 + 2 * j] == out[i, 2 + 2 * j], (i, cases[i])


def test_g4_onevar_qcqp_on_device(eng_mod, orc):
    """onevar_qcqp (utilities.py:241-288) on the device against the reference's 176 golden cases (every quirk of
    SURVEY.md appendix A): None / raise / point classification always; the point itself bit-exactly whenever the
    reference drew nothing from the RNG; for cases decided by a draw (zero objective, exact ties) the device's
    keyed draw must land in the feasible set the oracle computes."""
    from qcqp_amd import _ffi
    z = load_golden('g4_onevar_qcqp')
    N = len(z['s'])
    f0 = np.ascontiguousarray(z['f0'])
    fs = np.ascontiguousarray(z['fs'])
    nf = np.ascontiguousarray(z['nf'].astype(np.int32))
    s = np.ascontiguousarray(z['s'])
    x = np.zeros(N)
    st = np.zeros(N, dtype=np.int32)
    Cout = np.zeros((N, 5, 2))
    nC = np.zeros(N, dtype=np.int32)
    ip = C.POINTER(C.c_int)
    rc = _ffi.lib().qcqpmi_onevar_qcqp_batch(0, N, f0.ctypes.data_as(_ffi.c_dp), fs.ctypes.data_as(_ffi.c_dp),
                                              nf.ctypes.data_as(ip), s.ctypes.data_as(_ffi.c_dp), 123, x.ctypes.data_as(_ffi.c_dp),
                                              st.ctypes.data_as(ip), Cout.ctypes.data_as(_ffi.c_dp), nC.ctypes.data_as(ip))
    assert rc == 0
    exact = 0
    for i in range(N):
        if z['err'][i]:
            assert st[i] < 0, i
            continue
        if z['isnone'][i]:
            assert st[i] == 0, (i, st[i], x[i])
            continue
        assert st[i] == 1, (i, st[i])
        p0, q0 = z['f0'][i, 0], z['f0'][i, 1]
        if p0 == 0 and q0 == 0:
            # zero objective: uniform draw from a random interval (utilities.py:266-267); any point of the set is valid
            assert any(Cout[i, j, 0] <= x[i] <= Cout[i, j, 1] for j in range(nC[i])), (i, x[i])
        elif x[i] == z['x'][i]:
            exact += 1
        else:
            # only an exact tie between end points may differ (np.random.choice(bestxs), utilities.py:288)
            fv = lambda t: p0 * t * t + q0 * t
            assert z['draws'][i] > 0 and fv(x[i]) == fv(z['x'][i]), (i, x[i], z['x'][i])
            assert any(x[i] in (Cout[i, j, 0], Cout[i, j, 1]) for j in range(nC[i])), i
    assert exact >= 60


# ------------------------------------------------------------------ device-side ADMM setup (rocSOLVER)
@pytest.fixture(scope='module')
def rocsolver_loaded():
    """librocsolver.so is 0.9 GB: conftest.py starts reading it into the page cache when the GPU session starts; by the
    time this fixture runs (tests of this module come last) the dlopen is quick.  A box that has not delivered the file
    after another 6 minutes is too slow for this optional path: skip, with the reason."""
    from conftest import ROCSOLVER, rocsolver_warm
    if not rocsolver_warm(360.0):
        pytest.skip('librocsolver.so (0.9 GB) was not readable within the time budget on this box')
    C.CDLL(ROCSOLVER, mode=C.RTLD_GLOBAL)
    return True


def test_admm_device_eigh_rocsolver(eng_mod, orc, rocsolver_loaded):
    """qcqpmi_admm_setup: f.eigh of every constraint on the device (rocSOLVER batched dsyevd) instead of host
    LAPACK.  Rank-2 beamforming constraints have an (n-2)-dimensional null space: any orthonormal basis of it is
    a valid set of eigenvectors and the projections do not depend on the choice -- results agree with the
    host-eigh run and with the oracle to the accuracy the bisection (1e-6 on every multiplier) leaves after 80
    iterations."""
    from qcqp_amd import problems
    funcs, _, _ = problems.beamforming(12, 4, 3, seed=2)
    prob = orc.Problem(funcs)
    n, m = prob.n, prob.m
    rho = float(np.sqrt(m))
    P0 = np.asarray(funcs[0][0])
    Minv = np.linalg.inv(2. * (P0 + rho * m * np.eye(n)))
    R = 9
    X0 = np.random.RandomState(4).randn(n, R)
    res = []
    lm, Q = prob.eig()
    br = eng_mod.Engine.reference_bracket(lm)
    for device in (False, True):
        e = make(eng_mod, funcs)
        if device:
            e.admm_setup(method='rocsolver')
            # rocSOLVER's round-off eigenvalues of the null spaces differ from LAPACK's, and the reference's bracket is made
            # of them (utilities.py:176-180): start the bisections where the host run starts them
            e.admm_set_bracket(*br)
        else:
            e.admm_set_eig(lm, Q)
        e.upload(X0)
        proj = np.stack([e.admm_onecons(k) for k in range(1, m + 1)])      # onecons_qcqp(z, f_k): basis-invariant
        out = e.admm_run(rho, Minv, phase1=True, num_iters=80)
        res.append((e.download(), out, proj))
    (Xh, oh, ph), (Xd, od, pd) = res
    dp = np.max(np.abs(pd - ph)) / (1 + np.max(np.abs(ph)))
    print('\nrocSOLVER eigenpairs vs LAPACK: projections onecons_qcqp(z, f_k) differ by %.2e, 80 + 80 iterations by %.2e' % (dp, rel(Xd, Xh)))
    assert dp < 1e-9, dp
    assert rel(Xd, Xh) < 1e-6
    assert rel(od['f0'], oh['f0']) < 1e-6 and np.max(np.abs(od['maxviol'] - oh['maxviol'])) < 1e-6
    for r in range(3):
        xa = prob.improve_admm(X0[:, r], num_iters=80, rho=rho)
        assert rel(Xd[:, r], xa) < 1e-6, r


# ------------------------------------------------------------------ ADMM: unit operator golden + scale
def test_g5_onecons_on_device(eng_mod):
    """onecons_qcqp (utilities.py:290-315) on the device (secular-equation kernel of admm.h through
    qcqpmi_admm_onecons) against the reference's golden cases G5, fed the reference's eigenpairs: the projection
    of z onto {x : x^T P x + q^T x + r <= / == 0}, including the early returns (already feasible, P = 0)."""
    z = load_golden('g5_onecons')
    n = z['P'].shape[1]
    N = z['P'].shape[0]
    assert N >= 8
    worst = 0.0
    for i in range(N):
        funcs = [(np.eye(n), np.zeros(n), 0., None),
                 (z['P'][i], z['q'][i], float(z['r'][i]), RELSTR[int(z['relop'][i])])]
        e = make(eng_mod, funcs)
        e.admm_set_eig(z['lmb'][i][None], z['Q'][i][None])
        e.upload(np.stack([z['z'][i]] * 3, axis=1))      # three copies: every column must follow the reference
        x = e.admm_onecons(1)
        d = np.max(np.abs(x - z['x'][i][:, None]))
        worst = max(worst, d)
        assert d < 1e-9 * (1 + np.max(np.abs(z['x'][i]))), (i, bool(z['early'][i]), d)
    print('\ng5 onecons: %d cases, worst |dx| %.2e' % (N, worst))


def _eig_all(form):
    lm = np.zeros((form.m, form.n))
    Q = np.zeros((form.m, form.n, form.n))
    for k, f in enumerate(form.fs):
        lm[k], Q[k] = np.linalg.eigh(np.asarray(f.P.todense() if hasattr(f.P, 'todense') else f.P))
    return lm, Q


def _eig_lowrank(form, rank=2):
    """Full eigendecompositions of constraint matrices of rank <= `rank` in O(m n^2): range by a random probe, the small
    eigenproblem there, the null space completed by the Householder Q of a complete QR (80 LAPACK eigh calls at n = 1024
    cost minutes on a slow host; the pairs are checked against P Q = Q diag(lam) below)."""
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


@pytest.mark.parametrize('nant,mh,ml,R', [(64, 6, 10, 96), (512, 16, 64, 48)])
def test_admm_reduced_basis_vs_full_eigenbasis(eng_mod, orc, nant, mh, ml, R):
    """improve(ADMM) at the scale where the setup changes (BASELINE.json configs[3] family; second case = its full
    size n = 1024, m = 80): low-rank constraints take the reduced bases of lowrank.reduced_bases (randomised range
    finder + Jacobi, device products only, no host eigh) instead of m full eigendecompositions.  Both formulations
    are the same iteration in exact arithmetic: the consensus update z, every projection and the dual live in
    span(B_k) + the q direction.  Asserted: the reduced path against the full eigenbasis on the device restart by
    restart, and the full eigenbasis against the oracle (improve_admm) on two restarts at the smaller size."""
    from qcqp_amd import lowrank, problems
    from qcqp_amd.form import QCQPForm
    funcs, _, _ = problems.beamforming(nant, mh, ml, seed=1)
    form = QCQPForm.from_arrays(funcs)
    n, m = form.n, form.m
    rho, iters = 1.0, 30
    e = eng_mod.Engine(form)
    lam, Bv, qhat, info = lowrank.reduced_bases(e, form)
    assert info['rank'].max() <= 2      # |h^H x|^2 in real variables: rank 2
    e.admm_set_basis(lam, Bv, qhat)
    # the eigenpairs the full-eigenbasis run and the oracle get; the reduced path starts its bisections from THEIR bracket
    # (utilities.py:176-180: every eigenvalue counts, LAPACK's round-off eigenvalues of the null space included)
    lm, Q = _eig_all(form) if n <= 256 else _eig_lowrank(form)
    e.admm_set_bracket(*eng_mod.Engine.reference_bracket(lm))
    X0 = np.random.RandomState(7).randn(n, R)
    e.upload(X0)
    out = e.admm_run(rho, None, phase1=True, num_iters=iters)     # P0 = I: diagonal z-update on the device
    Xr = e.download()
    f0, mv = e.eval()
    assert rel(out['f0'], f0) < 1e-9 and np.max(np.abs(out['maxviol'] - mv)) < 1e-9
    e2 = eng_mod.Engine(form)
    e2.admm_set_eig(lm, Q)
    e2.upload(X0)
    out2 = e2.admm_run(rho, None, phase1=True, num_iters=iters)
    Xf = e2.download()
    d = np.max(np.abs(Xf - Xr), axis=0) / (1 + np.max(np.abs(Xf), axis=0))
    print('\nADMM n=%d m=%d R=%d: reduced vs full eigenbasis max|dx| median %.2e max %.2e; feasible %d / %d' % (
        n, m, R, np.median(d), d.max(), int((out['maxviol'] < 1e-2).sum()), int((out2['maxviol'] < 1e-2).sum())))
    assert np.median(d) < 1e-9 and d.max() < 1e-6, (np.median(d), d.max())
    assert rel(out['f0'], out2['f0']) < 1e-6
    assert np.array_equal(out['maxviol'] < 1e-2, out2['maxviol'] < 1e-2)
    if n <= 256:
        prob = orc.Problem(funcs)
        prob._eig = (np.ascontiguousarray(lm), np.ascontiguousarray(Q))
        for r, xa in zip((0, R - 1), oracle_map(lambda r: prob.improve_admm(X0[:, r], num_iters=iters, rho=rho), (0, R - 1))):
            assert rel(Xf[:, r], xa) < 1e-6, r
            assert rel(Xr[:, r], xa) < 1e-6, r      # the north-star tolerance, reduced bases included


@pytest.mark.parametrize('family', ['bls', 'dense', 'maxcut'])
def test_admm_setup_on_device_vs_lapack(eng_mod, orc, family):
    """What the reference asks LAPACK / SuperLU for in improve_admm (qcqp.py:224-227, 261-278), formed on the device:
    lambda_min(P0) by Lanczos with device products against numpy.linalg.eigvalsh, and (2 (P0 + rho m I))^-1 by
    Newton-Schulz on the engine's GEMM against numpy.linalg.inv THROUGH the iteration: the same ADMM run with the host
    inverse handed in and with the device-side one must end on the same points (1e-6, the north-star tolerance)."""
    from qcqp_amd import problems
    if family == 'bls':
        funcs = problems.boolean_least_squares(200, 50, seed=3)[0]       # P0 = A^T A: rank 50, lambda_min = 0
    elif family == 'dense':
        funcs = problems.dense_indefinite(96, 6, seed=5)[0]              # indefinite objective
    else:
        funcs = problems.maxcut(150, 0.5, seed=2)[0]
    e = make(eng_mod, funcs)
    P0 = np.asarray(funcs[0][0].todense() if hasattr(funcs[0][0], 'todense') else funcs[0][0], dtype=float)
    P0 = (P0 + P0.T) / 2.
    n, m = P0.shape[0], len(funcs) - 1
    ev = np.linalg.eigvalsh(P0)
    lmin, steps = e.p0_lambda_min()
    assert abs(lmin - ev[0]) <= 1e-9 * max(1.0, abs(ev).max()), (lmin, ev[0], steps)
    rho = (2. * (1. - ev[0]) / m if ev[0] < 0 else 1. / m) * 50.       # the reference's auto-rho (qcqp.py:272-278)
    res, its = e.admm_zsolver_device(rho)
    assert res < 1e-10, (res, its)
    Minv = np.linalg.inv(2. * (P0 + rho * m * np.eye(n)))
    lm, Q = orc.Problem(funcs).eig()
    X0 = np.random.RandomState(1).randn(n, 24)
    e.admm_set_eig(lm, Q)
    e.upload(X0)
    out_d = e.admm_run(rho, None, phase1=True, num_iters=25)           # device-side z-solver
    Xd = e.download()
    e.upload(X0)
    out_h = e.admm_run(rho, Minv, phase1=True, num_iters=25)           # host inverse, as in round 1
    Xh = e.download()
    assert rel(Xd, Xh) < 1e-6, rel(Xd, Xh)
    assert rel(out_d['f0'], out_h['f0']) < 1e-6
    assert np.array_equal(out_d['iters2'], out_h['iters2'])
    # the matrix is gone after a host-side call: Minv = None must be refused, not silently reuse something stale
    with pytest.raises(Exception):
        e.admm_run(rho, None, phase1=True, num_iters=1)


def test_cd_staged_run_two_contexts(eng_mod):
    """qcqpmi_cd_run_stage: the staged run (prepare / launch phase 2 / fetch) returns bit for bit what qcqpmi_cd_run
    returns, also when a second context prepares its population in between on its own stream -- the way bench.py overlaps
    step k + 1 with the phase-2 kernel of step k.  Stages out of order are refused."""
    from qcqp_amd import problems
    from qcqp_amd.engine import EngineError
    funcs, _, _ = problems.boolean_least_squares(256, 64, seed=2)
    e1, e2, e0 = make(eng_mod, funcs), make(eng_mod, funcs), make(eng_mod, funcs)
    R = 300
    ref = []
    for k in range(3):
        e0.randn(R, seed=50 + k, first_index=7)
        o = e0.cd_run(phase1=True, seed=50 + k, first_index=7)
        ref.append((o, e0.download(), e0.select_best(1e-4)))
    engs = [e1, e2]
    engs[0].randn(R, seed=50, first_index=7)
    engs[0].cd_begin(phase1=True, seed=50, first_index=7)
    for k in range(3):
        cur = engs[k % 2]
        cur.cd_phase2()
        if k + 1 < 3:
            nxt = engs[(k + 1) % 2]
            nxt.randn(R, seed=50 + k + 1, first_index=7)
            nxt.cd_begin(phase1=True, seed=50 + k + 1, first_index=7)
        o = cur.cd_fetch()
        X = cur.download()
        b = cur.select_best(1e-4)
        ro, rX, rb = ref[k]
        for key in ('sweeps1', 'sweeps2', 'visits2', 'accepted2', 'ran_phase2', 'f0', 'maxviol', 'status1', 'status2'):
            assert np.array_equal(o[key], ro[key]), (k, key)
        assert np.array_equal(X, rX)
        assert b[:3] == rb[:3] and np.array_equal(b[3], rb[3])
    with pytest.raises(EngineError):
        e1.cd_phase2()          # no stage 1 before it
    e1.randn(R, seed=1)
    e1.cd_begin(seed=1)
    with pytest.raises(EngineError):
        e1.cd_fetch()           # stage 2 skipped


# ------------------------------------------------------------------ the fused persistent ADMM kernel (round 3)
@pytest.mark.parametrize('nant,mh,ml,R,iters', [(64, 6, 10, 96, 60), (40, 3, 5, 7, 200), (512, 16, 64, 128, 40), (512, 16, 64, 1024, 25)])
def test_admm_fused_kernel_vs_multi_launch(eng_mod, orc, nant, mh, ml, R, iters):
    """improve_admm (qcqp.py:254-285) inside ONE persistent kernel per tile of restarts (csrc/admm_fused.hip: phase 1,
    better, phase 2 with bestx, better -- no host in the loop; a tile shared by a cluster of workgroups when there are few
    restarts) against the multi-launch path it replaces as the default (same iteration, other summation order in the two
    products): points, objective, max violation, iteration counts per restart.  Sizes: n = 128 (clusters of 4-16
    workgroups, ragged split of 16 constraints), n = 80 with 7 restarts (one partial tile), BASELINE.json configs[3]
    (n = 1024, m = 80) at the 8-GPU share of 128 restarts (clusters of 16) and at 1024 restarts (clusters of 4).  At the
    smaller sizes two restarts also go through the oracle (full eigenbasis there, reduced bases here: 1e-6)."""
    from qcqp_amd import lowrank, problems
    from qcqp_amd.form import QCQPForm
    funcs, _, _ = problems.beamforming(nant, mh, ml, seed=1)
    form = QCQPForm.from_arrays(funcs)
    n, m = form.n, form.m
    rho = 1.0
    e = eng_mod.Engine(form)
    lam, Bv, qhat, info = lowrank.reduced_bases(e, form)
    e.admm_set_basis(lam, Bv, qhat)
    # the oracle's eigenpairs (LAPACK at the small sizes) and the bracket the reference derives from them (utilities.py:176-180)
    lm, Q = _eig_all(form) if n <= 256 else _eig_lowrank(form)
    e.admm_set_bracket(*eng_mod.Engine.reference_bracket(lm))
    X0 = np.random.RandomState(7).randn(n, R)
    res = []
    for fused in (True, False, 2):        # 2: the fused kernel with four-wave workgroups, two per compute unit (round 5)
        e.admm_fused(fused)
        e.upload(X0)
        out = e.admm_run(rho, None, phase1=True, num_iters=iters)
        name, cw = e.last_admm_kernel()
        assert name == ('admm_fused_kernel' if fused else 'admm_multi_launch'), name
        res.append((e.download(), out, cw))
        f0, mv = e.eval()
        assert rel(out['f0'], f0) < 1e-9 and np.max(np.abs(out['maxviol'] - mv)) < 1e-9
    e.admm_fused(True)
    (Xf, of, cw), (Xm, om, _), (X4, o4, cw4) = res
    d4 = np.max(np.abs(X4 - Xm), axis=0) / (1 + np.max(np.abs(Xm), axis=0))
    print('\nfour-wave workgroups (clusters of %d) vs multi-launch: max|dx| median %.2e max %.2e' % (cw4, np.median(d4), d4.max()))
    assert np.median(d4) < 1e-9 and d4.max() < 1e-6 and rel(o4['f0'], om['f0']) < 1e-6
    assert np.mean((o4['iters1'] == om['iters1']) & (o4['iters2'] == om['iters2'])) > 0.97
    d = np.max(np.abs(Xf - Xm), axis=0) / (1 + np.max(np.abs(Xm), axis=0))
    same_it = np.mean((of['iters1'] == om['iters1']) & (of['iters2'] == om['iters2']))
    print('\nADMM fused (clusters of %d) vs multi-launch, n=%d m=%d R=%d: max|dx| median %.2e max %.2e; identical iteration counts '
          '%.1f %%; feasible %d / %d' % (cw, n, m, R, np.median(d), d.max(), 100 * same_it, int((of['maxviol'] < 1e-2).sum()),
                                         int((om['maxviol'] < 1e-2).sum())))
    assert cw >= 1
    assert np.median(d) < 1e-9 and d.max() < 1e-6, (np.median(d), d.max())
    assert rel(of['f0'], om['f0']) < 1e-6 and np.max(np.abs(of['maxviol'] - om['maxviol'])) < 1e-6
    assert same_it > 0.97
    # the fused kernel against the ORACLE (full eigenbasis there, reduced bases + the reference's bracket here) at every size:
    # the north-star tolerance on the points and on (objective, max violation)
    prob = orc.Problem(funcs)
    prob._eig = (np.ascontiguousarray(lm), np.ascontiguousarray(Q))
    sample = (0, R - 1)
    worst = 0.0
    for r, xa in zip(sample, oracle_map(lambda r: prob.improve_admm(X0[:, r], num_iters=iters, rho=rho), sample)):
        dd = rel(Xf[:, r], xa)
        worst = max(worst, dd)
        assert dd < 1e-6, (r, dd)
        fo, vo = prob.eval(0, xa), prob.max_violation(xa)
        assert abs(of['f0'][r] - fo) <= 1e-6 * (1 + abs(fo)), (r, of['f0'][r], fo)
        assert abs(of['maxviol'][r] - vo) <= 1e-6 * (1 + abs(vo)), (r, of['maxviol'][r], vo)
    print('fused kernel vs oracle (%d + %d iterations, restarts %s): worst %.2e' % (iters, iters, sample, worst))


def test_admm_fused_kernel_golden_and_phase2_only(eng_mod, orc):
    """The fused kernel on the reference's own golden run G8 (beamforming n = 40: improve_admm's result `xa` and its
    (f, v), 1e-6 -- the tolerance of the reference's bisection) through reduced bases of rank 2, and with phase1=False
    (qcqp.py:279-285) against the multi-launch path."""
    from conftest import funcs_from_npz
    from qcqp_amd import lowrank
    from qcqp_amd.form import QCQPForm
    z = load_golden('g8_admm_beam40')
    funcs = funcs_from_npz(z)
    form = QCQPForm.from_arrays(funcs)
    e = eng_mod.Engine(form)
    red = lowrank.reduced_bases(e, form, max_rank=2)
    assert red is not None
    lam, Bv, qhat, info = red
    e.admm_set_basis(lam, Bv, qhat)
    # the bracket the reference derived from ITS eigenvalues (stored in the fixture): LAPACK's round-off eigenvalues of the
    # 38-dimensional null spaces decide the end of the bracket of a rank-2 constraint (utilities.py:176-180)
    e.admm_set_bracket(*eng_mod.Engine.reference_bracket(z['lmb']))
    rho, iters = float(z['rho']), int(z['iters'])
    X0 = np.stack([z['x0']] * 5, axis=1)
    for p1 in (True, False):
        res = []
        for fused in (True, False):
            e.admm_fused(fused)
            e.upload(X0)
            out = e.admm_run(rho, None, phase1=p1, num_iters=iters)
            assert e.last_admm_kernel()[0] == ('admm_fused_kernel' if fused else 'admm_multi_launch')
            res.append((e.download(), out))
        assert rel(res[0][0], res[1][0]) < 1e-6
        assert rel(res[0][1]['f0'], res[1][1]['f0']) < 1e-6
        if p1:
            # against the reference's own result, through reduced bases and the fused kernel: the north-star tolerance
            worst = max(rel(res[0][0][:, r], z['xa']) for r in range(5))
            print('\nfused kernel, reduced bases vs the reference\'s improve_admm (G8 beam40): %.2e' % worst)
            assert worst < 1e-6, worst
            assert abs(res[0][1]['f0'][0] - z['fva'][0]) <= 1e-6 * (1 + abs(z['fva'][0]))
            assert abs(res[0][1]['maxviol'][0] - z['fva'][1]) <= 1e-6


def test_streaming_upload_of_coupled_constraints(eng_mod, orc, monkeypatch):
    """Coupled constraints beyond the budget for a row-major copy (16 GB; lowered to 100 kB here through QCQPMI_STREAM_LIMIT
    so that n = 100, m = 11 takes the path): every function is packed for the matrix cores when qcqpmi_set_quad receives it,
    dense or CSR, and nothing else is kept.  The packed problem must be the one the ordinary upload builds: evaluation of
    all functions and a coordinate-descent run bit for bit, and the oracle on a few restarts; separable problems are not
    affected by the limit (they keep their per-coordinate lists)."""
    import scipy.sparse as sp
    from qcqp_amd import problems
    n, m, R = 100, 11, 24
    funcs, _, _ = problems.dense_indefinite(n, m, seed=5)
    funcs = [(sp.csr_matrix(P) if k == 3 else P, q, r, rel) for k, (P, q, r, rel) in enumerate(funcs)]     # one function arrives as CSR
    X0 = 1.5 * np.random.RandomState(2).randn(n, R)
    e_ref = make(eng_mod, funcs)
    monkeypatch.setenv('QCQPMI_STREAM_LIMIT', '100000')
    e_str = make(eng_mod, funcs)
    bls = make(eng_mod, problems.boolean_least_squares(64, 16, seed=1)[0])
    monkeypatch.delenv('QCQPMI_STREAM_LIMIT')
    assert bls.separable
    res = []
    for e in (e_ref, e_str):
        f0, mv, F = e.eval_batch(X0, want_F=True)
        e.upload(X0)
        out = e.cd_run(phase1=True, num_iters=4, seed=7, first_index=3)
        assert e.last_cd_kernel() == 'dense_chain_mw_kernel'
        res.append((F, e.download(), out))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    for key in ('f0', 'maxviol', 'sweeps1', 'visits2', 'accepted2'):
        assert np.array_equal(res[0][2][key], res[1][2][key]), key
    prob = orc.Problem([(P.toarray() if sp.issparse(P) else P, q, r, rel) for (P, q, r, rel) in funcs])
    g0, gv, G = prob.eval_batch(X0, want_F=True)
    assert rel(res[1][0], G) < 1e-12
    # the reference-order mode needs the row-major matrices: refused on a streamed problem, with a message
    e_str.cd_reference_order(True)
    e_str.upload(X0)
    with pytest.raises(eng_mod.EngineError, match='reference-order'):
        e_str.cd_run(phase1=True, num_iters=1, seed=7)


def test_streaming_upload_with_single_coordinate_constraints_mixed_in(eng_mod, orc, monkeypatch):
    """Round-3 advisor finding: qcqpmi_set_quad streams only functions that couple coordinates; a box constraint x_i^2 <= c, a
    one-coordinate linear constraint or a constant one in a problem beyond the streaming limit was left without any packed
    copy and finalize read it from a null base.  Such functions are now packed at finalize from the host triplets.  Dense
    indefinite n = 100 with 9 coupled constraints + a box constraint (CSR), a one-coordinate linear constraint (dense) and a
    constraint without any coordinate: the streamed problem must be the ordinary one -- all function values and a
    coordinate-descent run bit for bit -- and agree with the oracle's evaluation."""
    import scipy.sparse as sp
    from qcqp_amd import problems
    n, R = 100, 24
    funcs, _, _ = problems.dense_indefinite(n, 9, seed=5)
    box = sp.csr_matrix(([1.0], ([17], [17])), shape=(n, n))
    qlin = np.zeros(n)
    qlin[40] = 2.0
    funcs = funcs[:4] + [(box, np.zeros(n), -9.0, '<=')] + funcs[4:7] + [(np.zeros((n, n)), qlin, -7.0, '<=')] + funcs[7:] + \
        [(np.zeros((n, n)), np.zeros(n), -1.0, '<=')]
    X0 = 1.5 * np.random.RandomState(2).randn(n, R)
    e_ref = make(eng_mod, funcs)
    monkeypatch.setenv('QCQPMI_STREAM_LIMIT', '100000')
    e_str = make(eng_mod, funcs)
    monkeypatch.delenv('QCQPMI_STREAM_LIMIT')
    res = []
    for e in (e_ref, e_str):
        f0, mv, F = e.eval_batch(X0, want_F=True)
        e.upload(X0)
        out = e.cd_run(phase1=True, num_iters=4, seed=7, first_index=3)
        assert e.last_cd_kernel() == 'dense_chain_mw_kernel'
        res.append((F, e.download(), out))
    assert np.array_equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1], res[1][1])
    for key in ('sweeps1', 'sweeps2', 'visits2', 'accepted2', 'f0', 'maxviol'):
        assert np.array_equal(res[0][2][key], res[1][2][key]), key
    prob = orc.Problem([(P.toarray() if sp.issparse(P) else P, q, r, rel) for (P, q, r, rel) in funcs])
    g0, gv, G = prob.eval_batch(X0, want_F=True)
    assert rel(res[1][0], G) < 1e-12


# ------------------------------------------------------------------ the chain kernels of the dense path (round 3)
@pytest.mark.parametrize('n,m,R,geom', [(250, 33, 100, (1, 64, 33, 64)),      # a free lane of the only wave is the serial thread
                                        (256, 64, 512, (1, 64, 64, 128)),     # full waves of constraints: an extra wave
                                        (1024, 256, 96, (1, 256, 256, 320)),  # BASELINE.json configs[4] family at the bench size
                                        (128, 600, 48, (2, 320, 320, 384)),   # two slots per thread (rows of G still in registers)
                                        (128, 1100, 32, (3, 384, 384, 448))]) # three slots: rows fetched visit by visit
def test_dense_chain_kernels_agree_bit_for_bit(eng_mod, n, m, R, geom):
    """dense_chain_mw_kernel (up to eight waves per restart, csrc/cd_dense_mw.h) against dense_chain_kernel (one wave per
    restart, the round-1/2 kernel): same expressions in the same order of operations per function, reductions over exact
    maxima / minima, a sorted gap list -- the points, values and counters must be IDENTICAL, phase 1 (bisection on the slack,
    qcqp.py:101-149) and phase 2 (qcqp.py:152-178).  `geom` = (slots per thread, constraint threads, serial thread, threads)
    documents the geometry each case exercises (mw_geometry)."""
    from qcqp_amd import problems
    form = problems.dense_indefinite_generated(n, m, seed=11)
    res = []
    for mode, name in ((0, 'dense_chain_mw_kernel'), (1, 'dense_chain_kernel')):
        e = eng_mod.Engine(form)
        e.dense_chain_mode(mode)
        e.randn(R, seed=3)
        out = e.cd_run(phase1=True, num_iters=3, seed=9)
        assert e.last_cd_kernel() == name
        res.append((e.download(), out))
    (X0, o0), (X1, o1) = res
    assert np.array_equal(X0, X1)
    for key in ('f0', 'maxviol', 'sweeps1', 'sweeps2', 'visits2', 'accepted2'):
        assert np.array_equal(o0[key], o1[key]), key
    assert o0['sweeps1'].sum() > 0
    if m < n:        # feasible families reach phase 2 (more constraints than variables: phase 1 only)
        assert o0['accepted2'].sum() > 0


def test_dense_chain_kernels_agree_with_equality_constraints(eng_mod, orc):
    """The same comparison on an UPLOADED problem with '==' constraints among the coupled ones (|f| <= s takes the general
    interval rule, utilities.py:209-231, in both kernels) -- and the points follow the oracle as far as the dense path can
    (same feasibility class; objective within the population's spread)."""
    from qcqp_amd import problems
    n, R = 96, 64
    funcs, _, _ = problems.dense_indefinite(n, 6, seed=5)
    rs = np.random.RandomState(2)
    # two equality constraints through the origin's neighbourhood: x' P x + q' x + r == 0 with an indefinite P
    for k in (2, 4):
        P, q, r, _ = funcs[k]
        funcs[k] = (P, q, 0.05 * rs.randn(), '==')
    res = []
    for mode, name in ((0, 'dense_chain_mw_kernel'), (1, 'dense_chain_kernel')):
        e = make(eng_mod, funcs)
        e.dense_chain_mode(mode)
        e.randn(R, seed=4)
        out = e.cd_run(phase1=True, num_iters=4, seed=8)
        assert e.last_cd_kernel() == name
        res.append((e.download(), out))
    (X0, o0), (X1, o1) = res
    assert np.array_equal(X0, X1)
    for key in ('f0', 'maxviol', 'sweeps1', 'sweeps2', 'visits2', 'accepted2'):
        assert np.array_equal(o0[key], o1[key]), key
    # reported values are those of the reported points (oracle evaluation)
    prob = orc.Problem(funcs)
    g0, gv = prob.eval_batch(X0)
    assert np.max(np.abs(g0 - o0['f0']) / (1.0 + np.abs(g0))) < 1e-9
    assert np.max(np.abs(gv - o0['maxviol']) / (1.0 + np.abs(gv))) < 1e-9
    assert o0['sweeps1'].sum() > 0


def test_dense_chain_falls_back_to_one_wave_beyond_eight_slots(eng_mod):
    """More than 8 constraints per thread of the multi-wave chain kernel (m > 3584) take dense_chain_kernel -- one wave per
    restart, four per-function arrays in LDS (120 KB at m = 3700) -- without being asked to: the run completes, the reported
    (objective, max violation) are those of the returned points, phase 1 reduces the violation."""
    from qcqp_amd import problems
    n, m, R = 128, 3700, 16
    form = problems.dense_indefinite_generated(n, m, seed=5)
    e = eng_mod.Engine(form)
    out4 = (C.c_int * 4)()
    assert e.L.qcqpmi_dense_chain_geometry(m, out4) == 0 and out4[0] > 8
    e.randn(R, seed=2)
    f_start, v_start = e.eval()
    out = e.cd_run(phase1=True, num_iters=2, seed=3)
    assert e.last_cd_kernel() == 'dense_chain_kernel'
    f0, mv = e.eval()
    assert np.max(np.abs(out['f0'] - f0) / (1.0 + np.abs(f0))) < 1e-9
    assert np.max(np.abs(out['maxviol'] - mv) / (1.0 + np.abs(mv))) < 1e-9
    assert out['sweeps1'].sum() > 0 and np.median(mv) < np.median(v_start)
