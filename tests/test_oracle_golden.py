"""Pins the CPU oracle (oracle/) to golden vectors captured from the reference itself
(tests/golden/*.npz, generator tools/gen_golden.py).  CPU-only."""
import os

import numpy as np
import pytest

from conftest import REPO, funcs_from_npz, load_golden, RELSTR

SMALL = ['bls10', 'bls32', 'maxcut12', 'dense16', 'beam10']


@pytest.mark.parametrize('name', SMALL)
def test_g1_eval_violation_better(orc, name):
    z = load_golden('g1_' + name)
    prob = orc.Problem(funcs_from_npz(z))
    X = z['X']
    f0, mv, F = prob.eval_batch(X, want_F=True)
    scale = 1 + np.abs(z['F'])
    assert np.max(np.abs(F - z['F']) / scale) < 1e-13
    assert np.max(np.abs(mv - z['maxviol']) / (1 + z['maxviol'])) < 1e-13
    for k in range(1, prob.m + 1):
        for s in range(X.shape[1]):
            assert abs(prob.violation(k, X[:, s]) - z['V'][k - 1, s]) <= 1e-13 * (1 + z['V'][k - 1, s])
    B = z['better']
    for a in range(8):
        for b in range(8):
            assert prob.better(X[:, a], X[:, b]) == B[a, b], (a, b)


@pytest.mark.parametrize('name', SMALL)
def test_g2_onevar_coeffs(orc, name):
    z = load_golden('g1_' + name)
    prob = orc.Problem(funcs_from_npz(z))
    for t in range(len(z['ov_k'])):
        for j in range(prob.m + 1):
            got = prob.onevar_coeffs(j, z['ov_x'][t], int(z['ov_k'][t]))
            ref = z['ov_T'][t, j]
            assert got[0] == ref[0]
            assert abs(got[1] - ref[1]) <= 1e-13 * (1 + abs(ref[1]))
            assert abs(got[2] - ref[2]) <= 1e-12 * (1 + abs(ref[2]))


def test_g3_feasible_intervals(orc):
    z = load_golden('g3_intervals')
    for row, out in zip(z['cases'], z['out']):
        p, q, r, s, rel = row
        I = orc.feasible_intervals(p, q, r, RELSTR[int(rel)], s)
        assert len(I) == int(out[0]), row
        for i, (lo, hi) in enumerate(I):
            # bit-exact: same IEEE operations in the same order
            assert lo == out[1 + 2 * i] and hi == out[2 + 2 * i], (row, I, out)


def test_g4_onevar_qcqp(orc):
    z = load_golden('g4_onevar_qcqp')
    for i in range(len(z['s'])):
        fs = [(f[0], f[1], f[2], RELSTR[int(f[3])]) for f in z['fs'][i][:int(z['nf'][i])]]
        np.random.seed(int(z['seed'][i]))
        rng = orc.Rng(orc.RNG_MT, 0, from_numpy_global=True)
        if z['err'][i]:
            with pytest.raises(RuntimeError):
                orc.onevar_qcqp(tuple(z['f0'][i]), fs, float(z['s'][i]), rng)
            continue
        x, C = orc.onevar_qcqp(tuple(z['f0'][i]), fs, float(z['s'][i]), rng)
        if z['isnone'][i]:
            assert x is None, i
        else:
            assert x is not None and x == z['x'][i], (i, x, z['x'][i])
        assert rng.draws == int(z['draws'][i]), i


def test_mt_stream_matches_numpy(orc):
    np.random.seed(123)
    rng = orc.Rng(orc.RNG_MT, 0, from_numpy_global=True)
    for k in [2, 3, 5, 7, 1, 8, 13, 2, 100]:
        assert rng.choice(k) == np.random.choice(k)
        assert rng.uniform(-1.5, 2.25) == np.random.uniform(-1.5, 2.25)
    np.random.seed(9)
    rng2 = orc.Rng(orc.RNG_MT, 9)
    assert rng2.uniform(0, 1) == np.random.uniform(0, 1)
    rng2.push_numpy()
    assert rng2.uniform(0, 1) == np.random.uniform(0, 1)


def test_g5_onecons(orc):
    z = load_golden('g5_onecons')
    n = z['P'].shape[1]
    for i in range(z['P'].shape[0]):
        funcs = [(np.eye(n), np.zeros(n), 0., None),
                 (z['P'][i], z['q'][i], float(z['r'][i]), RELSTR[int(z['relop'][i])])]
        prob = orc.Problem(funcs)
        x, steps = prob.onecons(1, z['z'][i], z['lmb'][i], z['Q'][i])
        assert (steps == -1) == bool(z['early'][i])
        assert np.max(np.abs(x - z['x'][i])) <= 1e-10 * (1 + np.max(np.abs(z['x'][i]))), i
        # own LAPACK eigh instead of the stored one: result is basis-invariant
        x2, _ = prob.onecons(1, z['z'][i])
        assert np.max(np.abs(x2 - z['x'][i])) <= 1e-7 * (1 + np.max(np.abs(z['x'][i]))), i


CD = ['bls10', 'bls32', 'bls64', 'maxcut12', 'dense16', 'dense32', 'circle5']


@pytest.mark.parametrize('name', CD)
def test_g6_cd_phase2(orc, name):
    z = load_golden('g6_cd_' + name)
    prob = orc.Problem(funcs_from_npz(z))
    for r in range(z['X0'].shape[1]):
        np.random.seed(int(z['p2_seed'][r]))
        x, stats = prob.cd_phase2(z['X0'][:, r])
        ref = z['p2_x'][:, r]
        assert np.max(np.abs(x - ref)) <= 1e-9 * (1 + np.max(np.abs(ref))), (name, r)
        f, v = prob.eval(0, x), prob.max_violation(x)
        assert abs(f - z['p2_fv'][r, 0]) <= 1e-9 * (1 + abs(f))
        assert abs(v - z['p2_fv'][r, 1]) <= 1e-9 * (1 + abs(v))


@pytest.mark.parametrize('name', CD)
def test_g7_cd_phase1_and_driver_mt(orc, name):
    """Phase 1 consumes the global MT19937 stream; the oracle replays it call-for-call."""
    z = load_golden('g6_cd_' + name)
    prob = orc.Problem(funcs_from_npz(z))
    iters = int(z['num_iters'])
    for r in range(z['Y0'].shape[1]):
        np.random.seed(int(z['seed0']) + r)
        rng = orc.Rng(orc.RNG_MT, 0, from_numpy_global=True)
        x, stats = prob.cd_phase1(z['Y0'][:, r], num_iters=iters, rng=rng)
        ref = z['p1_x'][:, r]
        assert np.max(np.abs(x - ref)) <= 1e-9 * (1 + np.max(np.abs(ref))), (name, r)
        rng.push_numpy()
        assert np.random.get_state()[2] == int(z['p1_pos'][r])
        np.random.seed(int(z['seed0']) + r)
        rng = orc.Rng(orc.RNG_MT, 0, from_numpy_global=True)
        x, s1, s2 = prob.improve_cd(z['Y0'][:, r], num_iters=iters, rng=rng)
        ref = z['full_x'][:, r]
        assert np.max(np.abs(x - ref)) <= 1e-9 * (1 + np.max(np.abs(ref))), (name, r)
        rng.push_numpy()
        assert np.random.get_state()[2] == int(z['full_pos'][r])


@pytest.mark.parametrize('name', ['beam10', 'beam40', 'bls10', 'dense16'])
def test_g8_admm(orc, name):
    z = load_golden('g8_admm_' + name)
    prob = orc.Problem(funcs_from_npz(z))
    prob._eig = (np.ascontiguousarray(z['lmb']), np.ascontiguousarray(z['Q']))
    iters = int(z['iters'])
    z1, _ = prob.admm_phase1(z['x0'], 1e-2, iters)
    assert np.max(np.abs(z1 - z['z1'])) <= 1e-7 * (1 + np.max(np.abs(z['z1'])))
    z2, _ = prob.admm_phase2(z['z1'], float(z['rho']), 1e-2, iters, 1e4)
    assert np.max(np.abs(z2 - z['z2'])) <= 1e-7 * (1 + np.max(np.abs(z['z2'])))
    rho = None if np.isnan(z['rho_arg']) else float(z['rho_arg'])
    xa = prob.improve_admm(z['x0'], num_iters=iters, rho=rho)
    assert np.max(np.abs(xa - z['xa'])) <= 1e-7 * (1 + np.max(np.abs(z['xa'])))
    assert abs(prob.eval(0, xa) - z['fva'][0]) <= 1e-7 * (1 + abs(z['fva'][0]))


@pytest.mark.parametrize('name', ['bls10', 'maxcut12'])
def test_g9_sdr_tail(orc, name):
    z = load_golden('g9_sdr_' + name)
    prob = orc.Problem(funcs_from_npz(z))
    mu, Sigma = orc.sdr_mu_sigma(z['X'], 1e-8, compat=True)
    assert np.array_equal(mu, z['mu'])
    assert np.max(np.abs(Sigma - z['Sigma'])) < 1e-15
    np.random.seed(int(z['seed']))
    for t in range(z['xs'].shape[1]):
        x, f, v = orc.suggest_sdr_sample(prob, mu, Sigma, bool(z['maximize']))
        assert np.max(np.abs(x - z['xs'][:, t])) < 1e-12
        assert abs(f - z['fv'][t, 0]) <= 1e-12 * (1 + abs(f))
        assert abs(v - z['fv'][t, 1]) <= 1e-12 * (1 + abs(v))


def test_oracle_under_sanitizers():
    """the CPU restatement runs clean under AddressSanitizer + UBSan (`make -C oracle asan`
    builds oracle/selftest.c against the oracle with -fsanitize=address,undefined)."""
    import subprocess
    odir = os.path.join(REPO, 'oracle')
    subprocess.check_call(['make', '-C', odir, 'asan'], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(odir, 'oracle_selftest_asan')], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         timeout=300)
    assert out.returncode == 0, out.stdout.decode()[-2000:]
    assert b'oracle selftest ok' in out.stdout


def test_incremental_cpu_baseline_matches_restatement(orc):
    """bench.py's optimised CPU baseline (incremental gradient) follows the faithful restatement's phase 2
    (same one-variable solver, different bookkeeping): same points to rounding, same visit / accept counts."""
    from qcqp_amd import problems
    for n, m_rows, seed in ((24, 16, 1), (64, 40, 2), (96, 30, 3)):
        funcs, _, _ = problems.boolean_least_squares(n, m_rows, seed=seed)
        prob = orc.Problem(funcs)
        for r in range(3):
            rng = orc.Rng(orc.RNG_KEYED, 5)
            rng.set_restart(r)
            x0 = orc.keyed_normal_matrix(5, n, 1, first_index=r)[:, 0]
            x1, _ = prob.cd_phase1(x0, num_iters=50, rng=rng)
            xa, sa = prob.cd_phase2(x1, rng=rng)
            xb, sb = prob.cd_phase2_incremental(x1, rng=rng)
            assert np.max(np.abs(xa - xb)) < 1e-9
            assert sa[1] == sb[1] and sa[2] == sb[2]


def test_generated_eval_matches_materialised_functions(orc):
    """orc_generated_eval (entry-by-entry values of a device-generated synthetic function, used by the full-size
    configs[4] test where no matrix can be materialised) against the same functions built as NumPy arrays from the same
    keyed stream (problems.materialise_generated)."""
    from qcqp_amd import problems
    n = 12
    form = problems.dense_indefinite_generated(n, 5, seed=7)
    funcs = problems.materialise_generated(form, orc.keyed_normal)
    X = np.random.RandomState(0).randn(n, 3)
    for k in range(6):
        P, q, r, _ = funcs[k]
        ref = np.array([X[:, s].dot(P).dot(X[:, s]) + q.dot(X[:, s]) + r for s in range(3)])
        got = orc.generated_eval(form.specs[k], k, n, X)
        assert np.allclose(ref, got, rtol=1e-13, atol=1e-13), (k, ref, got)


def test_g11_keyed_rng_branch_against_independent_philox_table(orc):
    """The KEYED branch of the oracle's random numbers -- orc_philox4x32, keyed_draw's counter layout, orc_rng_uniform /
    orc_rng_choice and orc_keyed_normal -- against tests/golden/g11_philox_keyed.npz: a table computed by
    tools/gen_philox_table.py with arbitrary-precision Python integers (nothing shared with the C code) and checked there
    against Random123's published known-answer vectors of philox4x32_10.  Every GPU parity test of phase 1 compares the
    engine with the oracle in this mode; no fixture of the reference can pin it (the reference draws from MT19937)."""
    import ctypes as C
    z = load_golden('g11_philox_keyed')
    L = orc.lib()
    out = (C.c_uint32 * 4)()
    for ctr, key, want in zip(z['kat_ctr'], z['kat_key'], z['kat_out']):      # Random123 known answers through the oracle's rounds
        L.orc_philox4x32((C.c_uint32 * 4)(*[int(v) for v in ctr]), (C.c_uint32 * 2)(*[int(v) for v in key]), out)
        assert [int(v) for v in out] == [int(v) for v in want]
    N = len(z['seed'])
    for j in range(N):
        sd, rr = int(z['seed'][j]), int(z['restart'][j])
        ctr = (C.c_uint32 * 4)(int(z['coord'][j]), int(z['sweep'][j]), int(z['it'][j]), rr & 0xffffffff)
        key = (C.c_uint32 * 2)(sd & 0xffffffff, (sd >> 32) & 0xffffffff)
        L.orc_philox4x32(ctr, key, out)
        assert [int(v) for v in out] == [int(v) for v in z['words'][j]], j
        g = orc.Rng(orc.RNG_KEYED, sd)
        g.set_restart(rr)
        g.set_ctx(z['coord'][j], z['sweep'][j], z['it'][j])
        assert g.uniform(float(z['lo'][j]), float(z['hi'][j])) == z['uniform'][j], j          # np.random.uniform stand-in (utilities.py:267)
        g.set_ctx(z['coord'][j], z['sweep'][j], z['it'][j])
        assert g.choice(int(z['k'][j])) == int(z['choice'][j]), j                              # np.random.choice stand-in (utilities.py:266, 288)
        assert g.choice(1) == 0
        nv = orc.keyed_normal(sd, rr, int(z['elem'][j]))
        assert abs(nv - z['normal'][j]) <= 4e-16 * (1.0 + abs(z['normal'][j])), (j, nv, z['normal'][j])   # libm's log / sin / cos: last-bit slack
