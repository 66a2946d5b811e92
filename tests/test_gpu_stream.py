"""GPU parity tests of the lifecycle launch (qcqpmi_cd_stream_run: cd_life_kernel, csrc/cd_life.hip -- since round 6 the ONE
lifecycle kernel; the round-4 kernel it replaced is a debug switch with one regression test in test_gpu_life.py): K populations of R
restarts -- suggest(RANDOM) + improve(COORD_DESCENT) + best point each, the reference's user loop (README.md:51-57, qcqp.py:381-382,
181-192) -- inside ONE persistent launch, against the serial path (one qcqpmi_pop_randn + qcqpmi_cd_run per population) and against
the oracle.  Run with `-m gpu` on an MI355X."""
import numpy as np
import pytest

from conftest import oracle_map

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng_mod():
    from qcqp_amd import engine
    assert engine.device_count() >= 1, 'no HIP device visible'
    return engine


LIFE = {'version': 0}
KNAME = {0: 'cd_life_kernel<3,band>', 2: 'cd_life_kernel<3,band>', 1: 'cd_phase2_qs_kernel<lifecycle>'}


@pytest.fixture(autouse=True, params=[0], ids=['cd_life_kernel'])      # (0 = the default dispatch: one kernel)
def life_version(request):
    LIFE['version'] = request.param
    yield request.param
    LIFE['version'] = 0


def make(eng_mod, funcs):
    from qcqp_amd.form import QCQPForm
    e = eng_mod.Engine(QCQPForm.from_arrays(funcs))
    e.cd_life_version(LIFE['version'])
    return e


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b)) / (1.0 + np.abs(np.asarray(b))))


COUNTERS = ('sweeps1', 'sweeps2', 'visits2', 'accepted2', 'ran_phase2', 'status1', 'status2')


@pytest.mark.parametrize('n,m_rows,R,K,iters', [(128, 32, 100, 3, 1000),      # R not a multiple of 16: populations straddle tiles
                                                 (48, 12, 20, 4, 3),           # sweep limit reached: frozen sweeps evaluate the objective
                                                 (1024, 256, 4096, 2, 1000),   # BASELINE.json configs[1], two steps in one launch
                                                 (256, 64, 5000, 1, 1000)])    # one population, more tiles than CUs
def test_cd_stream_run_equals_serial_runs(eng_mod, orc, n, m_rows, R, K, iters):
    """Every restart of a streamed run is the restart the serial path produces: same keyed normals, same phase-1 moves (the shared
    visit function cd_phase1_sep.h), same gate, the phase-2 arithmetic of the slot-queue kernel (a column's products depend on that
    column only) -- points IDENTICAL to rounding of nothing (asserted 1e-12), all counters equal; the reported objective is a fresh
    evaluation of the final point from the products of the restart's last sweep instead of the value tracked from an evaluated start
    (1e-11 relative), the max violation the same expression.  The best restart of every population (device selection) is the serial
    path's.  Two restarts per population also go through the ORACLE (improve_coord_descent with the same keyed stream): points 1e-9,
    objective and max violation against the oracle's evaluation of its own point 1e-9."""
    from qcqp_amd import problems
    funcs, _, _ = problems.boolean_least_squares(n, m_rows, seed=1)
    e = make(eng_mod, funcs)
    es = make(eng_mod, funcs)
    seed0, sstride, first0, fstride = 500, 3, 11, 70000
    if K > 1:
        es.cd_stream_reserve(K, R)          # buffers ahead of the run (qcqpmi_cd_stream_reserve): allocation only, same results
    o = es.cd_stream_run(K, R, num_iters=iters, seed=seed0, seed_stride=sstride, first_index=first0, first_stride=fstride)
    assert es.last_cd_kernel() == KNAME[LIFE['version']]
    assert es.pop_size == K * R
    X = es.download()
    f0e, mve = es.eval()        # fresh evaluation of all final points by the evaluation kernel
    assert rel(o['f0'], f0e) < 1e-11 and np.max(np.abs(o['maxviol'] - mve)) < 1e-12
    prob = orc.Problem(funcs)
    jobs = []
    for p in range(K):
        sd, fi = seed0 + p * sstride, first0 + p * fstride
        e.randn(R, seed=sd, first_index=fi)
        X0 = e.download()
        outr = e.cd_run(phase1=True, num_iters=iters, seed=sd, first_index=fi)
        Xr = e.download()
        sl = slice(p * R, (p + 1) * R)
        assert rel(X[:, sl], Xr) < 1e-12, (p, np.max(np.abs(X[:, sl] - Xr)))
        for key in COUNTERS:
            assert np.array_equal(o[key][sl], outr[key]), (p, key)
        assert rel(o['f0'][sl], outr['f0']) < 1e-11
        assert np.max(np.abs(o['maxviol'][sl] - outr['maxviol'])) < 1e-12
        idx, fb, vb, xb = e.select_best(1e-4)
        assert o['best_index'][p] == idx and o['best_f0'][p] == o['f0'][sl][idx] and o['best_maxviol'][p] == o['maxviol'][sl][idx]
        assert np.array_equal(o['best_x'][p], X[:, p * R + idx])
        jobs += [(p, r, sd, fi, X0[:, r].copy()) for r in (0, R - 1)]

    def oracle_restart(job):
        p, r, sd, fi, x0 = job
        rng = orc.Rng(orc.RNG_KEYED, sd)
        rng.set_restart(fi + r)
        return prob.improve_cd(x0, num_iters=iters, rng=rng)
    for (p, r, sd, fi, x0), (x, s1, s2) in zip(jobs, oracle_map(oracle_restart, jobs)):      # the oracle trajectories side by side
        if True:
            assert rel(X[:, p * R + r], x) < 1e-9, (p, r)
            # (a restart that phase 1 cannot improve any further stops after its first sweep without an update; the reference
            #  burns all num_iters sweeps on the same point: documented deviation 4)
            assert o['sweeps1'][p * R + r] == s1[0] or (s1[0] == iters and not o['ran_phase2'][p * R + r])
            assert o['visits2'][p * R + r] == s2[1] and o['accepted2'][p * R + r] == s2[2]
            assert abs(o['f0'][p * R + r] - prob.eval(0, x)) <= 1e-9 * (1 + abs(prob.eval(0, x)))
            assert abs(o['maxviol'][p * R + r] - prob.max_violation(x)) <= 1e-9


def test_cd_stream_run_scheduling_invariance(eng_mod):
    """Results do not depend on which slot, workgroup or episode a restart lands in: the same restarts as K = 1 population of
    4 R, as K = 4 populations of R with matching seeds / indices, and twice in a row -- bit for bit."""
    from qcqp_amd import problems
    funcs, _, _ = problems.boolean_least_squares(256, 64, seed=2)
    e = make(eng_mod, funcs)
    R = 1500
    a = e.cd_stream_run(1, 4 * R, seed=9, first_index=100)
    Xa = e.download()
    b = e.cd_stream_run(4, R, seed=9, seed_stride=0, first_index=100, first_stride=R)
    Xb = e.download()
    c = e.cd_stream_run(4, R, seed=9, seed_stride=0, first_index=100, first_stride=R)
    Xc = e.download()
    assert np.array_equal(Xa, Xb) and np.array_equal(Xb, Xc)
    for key in COUNTERS + ('f0', 'maxviol'):
        assert np.array_equal(a[key], b[key]) and np.array_equal(b[key], c[key]), key


def test_cd_stream_run_uploaded_starts_and_gate(eng_mod, orc):
    """generate = 0: the resident points are the starts (improve() on points the user set); phase1 = False (qcqp.py:186-192):
    starts within the slack of +-1 pass the gate of improve_coord_descent and run phase 2, random starts do not -- their
    phase 2 does not run, the point stays, and the reported (objective, max violation) is the evaluation of the start (one
    frozen sweep inside the kernel).  Against the serial path (identical points and counters) and the evaluation kernel."""
    from qcqp_amd import problems
    n, R = 128, 90
    funcs, _, _ = problems.boolean_least_squares(n, 40, seed=3)
    e = make(eng_mod, funcs)
    rs = np.random.RandomState(5)
    X0 = np.sign(rs.randn(n, R)) * (1.0 + 2e-5 * rs.rand(n, R))
    X0[:, ::3] = rs.randn(n, len(range(0, R, 3)))           # every third start is infeasible
    e.upload(X0)
    outr = e.cd_run(phase1=False, num_iters=1000, seed=4, first_index=9)
    Xr = e.download()
    e.upload(X0)
    o = e.cd_stream_run(1, R, generate=False, phase1=False, num_iters=1000, seed=4, first_index=9)
    X = e.download()
    assert rel(X, Xr) < 1e-12
    for key in COUNTERS:
        assert np.array_equal(o[key], outr[key]), key
    assert np.array_equal(o['ran_phase2'][::3], np.zeros(len(range(0, R, 3)), dtype=np.uint8)) and o['ran_phase2'][1] == 1
    assert np.array_equal(X[:, ::3], X0[:, ::3])
    f0e, mve = e.eval()
    assert rel(o['f0'], f0e) < 1e-11 and np.max(np.abs(o['maxviol'] - mve)) < 1e-12
    assert rel(o['f0'], outr['f0']) < 1e-11
    with pytest.raises(eng_mod.EngineError):
        e.cd_stream_run(2, R, generate=False)            # the resident population is not 2 R points


def test_cd_stream_run_exact_ties_take_the_reference_path(eng_mod, orc):
    """Objective x'x with x_i^2 == 1 at n = 48: the vertex of every scalar problem is exactly 0, the midpoint between the two
    feasible intervals -- EVERY visit is a tie that the kernel hands to the loop in the reference's arithmetic (keyed
    np.random.choice stand-in), block after block; in lifecycle mode that loop works on the objective RELATIVE to the start of
    phase 2 and keeps the window sum of the final evaluation.  Against the oracle restart by restart."""
    n, R = 48, 23
    funcs = [(np.eye(n), np.zeros(n), 0.0, None)]
    for i in range(n):
        P = np.zeros((n, n))
        P[i, i] = 1.0
        funcs.append((P, np.zeros(n), -1.0, '=='))
    e = make(eng_mod, funcs)
    prob = orc.Problem(funcs)
    rs = np.random.RandomState(n)
    X0 = np.sign(rs.randn(n, R)) * (1.0 + 2e-3 * rs.rand(n, R))
    seed, first = 77, 2
    e.upload(X0)
    o = e.cd_stream_run(1, R, generate=False, phase1=False, num_iters=30, seed=seed, first_index=first)
    X = e.download()
    for r in range(R):
        rng = orc.Rng(orc.RNG_KEYED, seed)
        rng.set_restart(first + r)
        x, s1, s2 = prob.improve_cd(X0[:, r], num_iters=30, phase1=False, rng=rng)
        assert rel(X[:, r], x) < 1e-12, (r, np.max(np.abs(X[:, r] - x)))
        assert o['visits2'][r] == s2[1] and o['accepted2'][r] == s2[2], r
        assert abs(o['f0'][r] - prob.eval(0, x)) <= 1e-11 * (1 + abs(prob.eval(0, x)))


def test_cd_stream_run_refuses_other_families(eng_mod):
    """No silent fallback: a family the kernel does not take is refused with a message that names the serial entry point -- for the
    round-4 kernel anything but the Boolean family with n a multiple of 16; for cd_life_kernel more than FOUR constraint classes (a
    different right-hand side on six kinds of coordinates), which it refuses, while n = 40 runs and two classes run (round 6:
    the multi-class kind)."""
    from qcqp_amd import problems
    funcs, _, _ = problems.boolean_least_squares(40, 10, seed=1)       # n not a multiple of 16
    e = make(eng_mod, funcs)
    if LIFE['version'] == 1:
        with pytest.raises(eng_mod.EngineError, match='lifecycle'):
            e.cd_stream_run(2, 32)
    else:
        e.cd_stream_run(2, 32)
        assert e.last_cd_kernel() == 'cd_life_kernel<3,band>'
    funcs2 = [funcs[0]] + [(P * (1.0 + (i % 2)), q, r * (1.0 + (i % 2)), rl) for i, (P, q, r, rl) in enumerate(funcs[1:])]
    e2 = make(eng_mod, funcs2)
    if LIFE['version'] != 1:
        e2.cd_stream_run(2, 32)
        assert e2.last_cd_kernel() == 'cd_life_kernel<3,gen,classes>'
    funcs6 = [funcs[0]] + [(P * (1.0 + (i % 6)), q, r * (1.0 + (i % 6)), rl) for i, (P, q, r, rl) in enumerate(funcs[1:])]
    e6 = make(eng_mod, funcs6)
    with pytest.raises(eng_mod.EngineError, match='lifecycle') as ei:
        e6.cd_stream_run(2, 32)
    assert ei.value.code == eng_mod.E_UNSUPPORTED


@pytest.mark.parametrize('K,R,iters,p1', [(3, 1, 1000, True),       # single-restart populations (the reference's own use)
                                           (1, 5, 1000, True),       # fewer restarts than slots of one workgroup
                                           (2, 33, 0, True),         # num_iters = 0: no sweep of either phase
                                           (2, 40, 50, False)])      # random starts without phase 1: nothing passes the gate
def test_cd_stream_run_edge_shapes(eng_mod, K, R, iters, p1):
    """Degenerate shapes of a streamed run against the serial path: populations of one restart, fewer restarts than a workgroup has
    slots, num_iters = 0 (qcqp.py:110, 160: no sweep at all -- the result is the start, evaluated), phase1 = False on random starts
    (qcqp.py:186-189: the gate keeps every restart out of phase 2; objective and max violation of the untouched start come from the
    kernel's frozen sweep)."""
    from qcqp_amd import problems
    n = 48
    funcs, _, _ = problems.boolean_least_squares(n, 16, seed=4)
    e = make(eng_mod, funcs)
    es = make(eng_mod, funcs)
    o = es.cd_stream_run(K, R, phase1=p1, num_iters=iters, seed=21, seed_stride=5, first_index=3, first_stride=1000)
    X = es.download()
    f0e, mve = es.eval()
    assert rel(o['f0'], f0e) < 1e-11 and np.max(np.abs(o['maxviol'] - mve)) < 1e-12
    for p in range(K):
        sd, fi = 21 + 5 * p, 3 + 1000 * p
        e.randn(R, seed=sd, first_index=fi)
        X0 = e.download()
        outr = e.cd_run(phase1=p1, num_iters=iters, seed=sd, first_index=fi)
        Xr = e.download()
        sl = slice(p * R, (p + 1) * R)
        assert rel(X[:, sl], Xr) < 1e-12, p
        for key in COUNTERS:
            assert np.array_equal(o[key][sl], outr[key]), (p, key)
        assert rel(o['f0'][sl], outr['f0']) < 1e-11
        if iters == 0 or not p1:
            assert np.array_equal(X[:, sl], X0) and not o['sweeps2'][sl].any()
        idx = e.select_best(1e-4)[0]
        assert o['best_index'][p] == idx


def test_streamed_run_exchange_over_rccl_one_rank(eng_mod):
    """The exchange of a streamed run through the library's RCCL communicator (one rank: the only size a one-GPU box offers --
    the N-rank logic is the CPU test test_global_best_of_populations_two_ranks): dist.global_best_of_populations with
    Engine.comm_allreduce as its all-reduce returns the local winners unchanged, and qcqpmi_comm_allreduce moves tables far
    beyond the four scalars it was limited to (keys of all populations, K x n points)."""
    from qcqp_amd import dist, problems
    funcs, _, _ = problems.boolean_least_squares(64, 16, seed=7)
    e = make(eng_mod, funcs)
    dist.init_rccl(e, 0, 1)
    K, R = 6, 50
    o = e.cd_stream_run(K, R, seed=3, seed_stride=1)
    big = np.arange(20000, dtype=np.float64)
    assert np.array_equal(e.comm_allreduce(big.copy(), 'sum'), big) and np.array_equal(e.comm_allreduce(big.copy(), 'max'), big)
    calls = []

    def allreduce(a):
        calls.append(a.size)
        return e.comm_allreduce(a, 'sum')
    # world = 1 skips the exchange; force the two all-reduces through RCCL by calling them the way a rank of a larger job does
    keys = np.zeros((1, K, 3))
    keys[0, :, 0], keys[0, :, 1], keys[0, :, 2] = o['best_f0'], o['best_maxviol'], o['best_index']
    kk = allreduce(keys.ravel().copy()).reshape(1, K, 3)
    xx = allreduce(o['best_x'].ravel().copy()).reshape(K, -1)
    assert calls == [K * 3, K * 64] and np.array_equal(kk, keys) and np.array_equal(xx, o['best_x'])
    ks, X = dist.global_best_of_populations(allreduce, 0, 1, o['best_f0'], o['best_maxviol'], o['best_index'], o['best_x'])
    assert [k[0] for k in ks] == list(o['best_index']) and np.array_equal(X, o['best_x'])

