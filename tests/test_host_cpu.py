"""CPU-only tests of the host side: C-ABI symbol export, problem containers, sharding and the
selection rule (incl. a 2-process gloo run).  No GPU compute is called here."""
import ctypes
import os
import re
import subprocess
import sys
import time

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from qcqp_amd import _build, _ffi
    _build.build()
    hdr = open(os.path.join(REPO, 'include', 'qcqp_mi.h')).read()
    declared = set(re.findall(r'\b(qcqpmi_[a-z0-9_]+)\s*\(', hdr))
    declared.discard('qcqpmi_ctx')
    bound = set(name for name, _, _ in _ffi.PROTOTYPES)
    assert declared == bound, (declared ^ bound)
    lib = ctypes.CDLL(_ffi.LIBPATH)
    for name in declared:
        assert hasattr(lib, name), name
    L = _ffi.lib()
    assert L.qcqpmi_abi_version() == 6
    assert L.qcqpmi_device_count() >= 0


def test_abi_argument_checks_need_no_gpu():
    """Every entry point refuses a NULL context / bad arguments with QCQPMI_EINVAL before it touches a device: the error
    behaviour of the boundary is testable on the build host (no compute call is made)."""
    from qcqp_amd import _ffi
    L = _ffi.lib()
    EINVAL = -1
    z = np.zeros(4)
    zi = np.zeros(4, dtype=np.int64)
    dp = z.ctypes.data_as(_ffi.c_dp)
    ip = zi.ctypes.data_as(_ffi.c_ip)
    assert L.qcqpmi_admm_zsolver_device(None, 1.0, 0, dp, ip) == EINVAL
    assert L.qcqpmi_p0_lambda_min(None, 0, 1e-12, dp, ip) == EINVAL
    assert L.qcqpmi_debug_trace(None, ip, 4) == EINVAL
    assert L.qcqpmi_debug_profile(None, 0, None) == EINVAL
    assert L.qcqpmi_admm_run(None, 1, 10, 1e-2, 1e4, 1.0, None, None, None, None, None) == EINVAL
    assert L.qcqpmi_pop_sdr_sample(None, None, None, 4, 0, 0, None) != 0
    assert (L.qcqpmi_last_cd_kernel(None) or b'') == b''
    assert L.qcqpmi_sync(None) == EINVAL
    # a context on a machine without a GPU cannot be created: the error is reported, not a crash
    h = ctypes.c_void_p()
    if L.qcqpmi_device_count() == 0:
        assert L.qcqpmi_ctx_create(ctypes.byref(h), 4, 1, 0) != 0


def test_engine_fails_loudly_without_gpu():
    from qcqp_amd import _ffi, problems
    from qcqp_amd.form import QCQPForm
    if _ffi.lib().qcqpmi_device_count() > 0:
        pytest.skip('a GPU is visible')
    from qcqp_amd.engine import Engine, EngineError
    funcs, _, _ = problems.boolean_least_squares(6, 8, seed=1)
    with pytest.raises(EngineError) as ei:
        Engine(QCQPForm.from_arrays(funcs))
    assert 'no CPU fallback' in str(ei.value)


def test_form_mirrors_reference_fields():
    from qcqp_amd import problems, settings
    from qcqp_amd.form import QCQPForm
    funcs, maxi, _ = problems.maxcut(9, 0.5, seed=4)
    form = QCQPForm.from_arrays(funcs)
    assert form.n == 9 and form.m == 9 and maxi
    assert form.f0.relop is None and all(f.relop == '==' for f in form.fs)
    assert form.fi(2) is form.fs[2]
    assert np.allclose(form.f0.P, form.f0.P.T)
    assert settings.improve_methods == ['coord-descent', 'admm', 'dccp', 'ipopt']
    assert settings.suggest_methods == ['random', 'sdr', 'spectral']
    import qcqp_amd
    assert qcqp_amd.COORD_DESCENT == 'coord-descent' and qcqp_amd.SDR == 'sdr'


def test_problem_generators_match_example_scripts():
    from qcqp_amd import problems
    funcs, _, extra = problems.boolean_least_squares(10, 15, seed=1, legacy_seed=True)
    np.random.seed(1)
    A = np.random.randn(15, 10)
    b = np.random.randn(15, 1)
    assert np.array_equal(extra['A'], A) and np.array_equal(extra['b'], b)
    assert np.allclose(funcs[0][0], A.T.dot(A)) and np.allclose(funcs[0][1], (-2 * A.T.dot(b)).ravel())
    funcs, _, _ = problems.beamforming(4, 3, 2, seed=1)
    assert len(funcs) == 6 and funcs[1][3] == '<=' and funcs[1][2] == 20.0 and funcs[5][2] == -2.0


def test_shard_range_partitions_global_indices():
    from qcqp_amd import dist
    for total in (1, 7, 4096, 8191):
        for world in (1, 2, 3, 8):
            seen = []
            for rank in range(world):
                first, cnt = dist.shard_range(total, rank, world)
                seen.extend(range(first, first + cnt))
            assert seen == list(range(total))


def test_selection_rule_matches_better_fold(orc):
    """better_key ordering == folding QCQPForm.better over the candidates (ties -> lowest index)."""
    from conftest import funcs_from_npz, load_golden
    from qcqp_amd import dist
    z = load_golden('g1_bls10')
    prob = orc.Problem(funcs_from_npz(z))
    X = z['X']
    f0, mv = prob.eval_batch(X)
    key = dist.select_best_host(f0, mv, 1e-4)
    best = 0
    for sidx in range(1, X.shape[1]):
        a = (int(mv[sidx] / 1e-4), f0[sidx])
        b = (int(mv[best] / 1e-4), f0[best])
        if a < b:
            best = sidx
    assert key[2] == best
    # and agrees with the reference's pairwise rule on every pair of distinct keys
    for a in range(X.shape[1]):
        for b in range(X.shape[1]):
            ka, kb = dist.better_key(f0[a], mv[a], a), dist.better_key(f0[b], mv[b], b)
            if ka[:2] != kb[:2]:
                assert (prob.better(X[:, a], X[:, b]) == 1) == (ka < kb)


WORKER = r'''
import os, sys
sys.path.insert(0, %(repo)r)
import numpy as np
from qcqp_amd import dist, problems
from oracle import oracle as orc
kids = dist.spawn_local_ranks(int(os.environ.get('TEST_SELF_SPAWN', '1')))   # no-op under a launcher
rank, local, world = dist.env_world()


class GlooTransport(object):
    # same three calls as dist.FileRendezvous, on torch.distributed's gloo backend (test-side only)
    def __init__(self):
        import torch.distributed as td
        self.td = td
        td.init_process_group(backend='gloo')
        self.rank, self.world = td.get_rank(), td.get_world_size()
    def broadcast_bytes(self, payload, src=0):
        obj = [payload if self.rank == src else None]
        self.td.broadcast_object_list(obj, src=src)
        return obj[0]
    def allgather(self, obj):
        out = [None] * self.world
        self.td.all_gather_object(out, obj)
        return out
    def barrier(self):
        self.td.barrier()
    def close(self):
        self.td.barrier()


boot = GlooTransport() if os.environ.get('TEST_TRANSPORT') == 'gloo' else dist.FileRendezvous(rank, world)
funcs, _, _ = problems.boolean_least_squares(12, 16, seed=5)
prob = orc.Problem(funcs)
R = 10                                  # global restarts, sharded by index
first, cnt = dist.shard_range(R, rank, world)
X = np.stack([orc.keyed_normal_matrix(77, 12, 1, first_index=first + r)[:, 0] for r in range(cnt)], axis=1)
out = []
for r in range(cnt):
    rng = orc.Rng(orc.RNG_KEYED, 77); rng.set_restart(first + r)
    x, _, _ = prob.improve_cd(X[:, r], num_iters=50, rng=rng)
    out.append(x)
out = np.stack(out, axis=1)
f0, mv = prob.eval_batch(out)
key = dist.select_best_host(f0, mv, 1e-4, index_offset=first)
gkey, gx = dist.global_best_host(boot, key, out[:, key[2] - first])
np.save(os.path.join(%(tmp)r, 'rank%%d.npy' %% rank), np.concatenate([[gkey[0], gkey[1], gkey[2], world], gx]))
boot.close()
sys.exit(dist.wait_children(kids))
'''


def _check_two_rank_result(tmp_path, orc):
    from qcqp_amd import dist, problems
    r0 = np.load(tmp_path / 'rank0.npy')
    r1 = np.load(tmp_path / 'rank1.npy')
    assert np.array_equal(r0, r1) and int(r0[3]) == 2
    # single-process reference
    funcs, _, _ = problems.boolean_least_squares(12, 16, seed=5)
    prob = orc.Problem(funcs)
    xs = []
    for r in range(10):
        rng = orc.Rng(orc.RNG_KEYED, 77)
        rng.set_restart(r)
        x0 = orc.keyed_normal_matrix(77, 12, 1, first_index=r)[:, 0]
        xs.append(prob.improve_cd(x0, num_iters=50, rng=rng)[0])
    xs = np.stack(xs, axis=1)
    f0, mv = prob.eval_batch(xs)
    key = dist.select_best_host(f0, mv, 1e-4)
    assert int(r0[2]) == key[2] and r0[1] == key[1]
    assert np.array_equal(r0[4:], xs[:, key[2]])


@pytest.mark.parametrize('transport', ['file', 'gloo'])
def test_two_process_launcher_selection(tmp_path, orc, transport):
    '''world_size 2 on CPU under torch.distributed.run (the driver's launcher): restarts sharded by global
    index, the final exchange picks the same global best on both ranks, identical to the single-process
    answer.  transport=file is the product's own rendezvous (qcqp_amd.dist.FileRendezvous, torch-free);
    transport=gloo runs the same exchange on torch.distributed's gloo backend.'''
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(repo=REPO, tmp=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', TEST_TRANSPORT=transport)
    env.pop('WORLD_SIZE', None); env.pop('RANK', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', '29577' if transport == 'file' else '29578', str(script)]
    subprocess.run(cmd, check=True, env=env, timeout=600, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    _check_two_rank_result(tmp_path, orc)


def test_two_process_self_spawn_selection(tmp_path, orc):
    '''Plain `python worker.py` with no launcher: rank 0 spawns rank 1 itself (dist.spawn_local_ranks, the
    path `python bench.py --gpus N` takes) and both meet through the file rendezvous.'''
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(repo=REPO, tmp=str(tmp_path)))
    env = dict(os.environ, TEST_SELF_SPAWN='2')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'QCQP_AMD_RDZV'):
        env.pop(k, None)
    subprocess.run([sys.executable, str(script)], check=True, env=env, timeout=600, stdout=subprocess.PIPE,
                   stderr=subprocess.STDOUT)
    _check_two_rank_result(tmp_path, orc)


def test_rendezvous_rejects_and_clears_leftovers_of_a_dead_job(tmp_path, monkeypatch):
    """External launcher: the rendezvous key is predictable (launcher pid + endpoint), so a directory with files of a dead job
    can pre-exist.  A message file older than the job's epoch is not a message; rank 0 removes such files when it opens the
    directory; the epoch is the later of the launcher's start and this rank's start minus the window in which the ranks of one
    job start; without a launcher start time (no /proc entry) a warning is raised and the own start time alone decides."""
    import warnings
    from qcqp_amd import dist
    monkeypatch.delenv(dist.RDZV_ENV, raising=False)
    monkeypatch.setenv('XDG_RUNTIME_DIR', str(tmp_path))
    monkeypatch.setenv('MASTER_PORT', '29999')
    now = time.time()
    ep = dist._job_epoch()
    assert now - dist.RANK_START_WINDOW - 5.0 <= ep <= now
    # a stale message of the same key, 1000 s old: rejected by _fresh, removed by rank 0
    key = dist._job_key()
    d = os.path.join(str(tmp_path), 'qcqp_amd_rdzv', key)
    os.makedirs(d, mode=0o700)
    os.chmod(os.path.join(str(tmp_path), 'qcqp_amd_rdzv'), 0o700)
    stale = os.path.join(d, 'b000001')
    with open(stale, 'wb') as f:
        f.write(b'Jnull')
    os.utime(stale, (now - 1000.0, now - 1000.0))
    cls = dist.FileRendezvous if hasattr(dist, 'FileRendezvous') else dist.FileComm
    c1 = cls(rank=1, world=2)
    assert not c1._fresh(stale) and os.path.exists(stale)           # rank 1 ignores it ...
    c0 = cls(rank=0, world=2)
    assert not os.path.exists(stale)                                # ... rank 0 clears it
    c0._put('b000001', b'fresh')
    assert c1._fresh(os.path.join(d, 'b000001'))
    # launcher start time unavailable: a warning, and the own start decides
    monkeypatch.setattr(dist, '_proc_start_time', lambda pid: 0.0)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        ep2 = dist._job_epoch()
    assert any('start time of the launcher' in str(x.message) for x in w)
    assert time.time() - dist.RANK_START_WINDOW - 5.0 <= ep2 <= time.time()


def test_product_package_does_not_import_torch():
    '''The north star: no PyTorch anywhere in the product (torch stays test/launcher-side plumbing).'''
    import re
    pkg = os.path.join(REPO, 'qcqp_amd')
    for name in os.listdir(pkg):
        if name.endswith('.py'):
            src = open(os.path.join(pkg, name)).read()
            assert not re.search(r'^\s*(import|from)\s+torch', src, re.M), name


# ------------------------------------------------------------------ host logic of the SDP relaxation
def test_sdr_family_detection_and_lifted_cost():
    """qcqp_amd/sdr.py (no GPU): the unit-diagonal family is recognised (dense and sparse constraint
    matrices, any positive d_i), everything else is refused, and the lifted cost reproduces the objective:
    f0(x) = [y; 1]' C [y; 1] with x = sqrt(d) * y."""
    import scipy.sparse as sp
    from qcqp_amd import problems, sdr
    from qcqp_amd.form import QCQPForm
    funcs, _, _ = problems.boolean_least_squares(7, 9, seed=3)
    form = QCQPForm.from_arrays(funcs)
    d = sdr.unit_diagonal_family(form)
    assert d is not None and np.allclose(d, 1.0)
    # x_i^2 = d_i with different d, sparse matrices
    n = 5
    rs = np.random.RandomState(0)
    G = rs.randn(n, n)
    dd = np.array([0.5, 2.0, 1.0, 4.0, 0.25])
    fs = [((G + G.T) / 2, rs.randn(n), 0.3, None)]
    for i in range(n):
        P = sp.csr_matrix(([3.0], ([i], [i])), shape=(n, n))
        fs.append((P, np.zeros(n), -3.0 * dd[i], '=='))
    form2 = QCQPForm.from_arrays(fs)
    d2 = sdr.unit_diagonal_family(form2)
    assert np.allclose(d2, dd)
    C, sc = sdr.lifted_cost(form2, d2)
    assert np.allclose(C, C.T) and np.allclose(sc, np.append(np.sqrt(dd), 1.0))
    y = rs.randn(n)
    x = np.sqrt(dd) * y
    P0, q0, r0 = fs[0][0], fs[0][1], fs[0][2]
    z = np.append(y, 1.0)
    assert abs(z.dot(C.dot(z)) - (x.dot(P0.dot(x)) + q0.dot(x) + r0)) < 1e-12
    # refused: inequality, linear term, off-diagonal entry, a coordinate without constraint, d <= 0
    bad = [fs[:1] + [(np.eye(n), np.zeros(n), -1.0, '<=')] * n,
           fs[:1] + [(np.diag(np.eye(n)[i]), np.eye(n)[i], -1.0, '==') for i in range(n)],
           fs[:1] + [(np.ones((n, n)), np.zeros(n), -1.0, '==')] * n,
           fs[:-1],
           fs[:1] + [(np.diag(np.eye(n)[i]), np.zeros(n), +1.0, '==') for i in range(n)]]
    for b in bad:
        assert sdr.unit_diagonal_family(QCQPForm.from_arrays(b)) is None


def test_sdr_dual_certificate_on_a_known_solution():
    """C = -J (all ones) has the SDP optimum X = J (rank one, v_i identical): y_i = N - ... certificate PSD,
    zero gap; a non-optimal V gives a negative lambda_min and a bound below the optimum."""
    from qcqp_amd import sdr
    N = 6
    C = -np.ones((N, N))
    V = np.zeros((N, 64)); V[:, 0] = 1.0            # X = J
    y, lmin, lower = sdr.dual_certificate(C, V)
    assert np.allclose(y, N) and lmin > -1e-12
    assert abs(lower - (-N * N)) < 1e-9              # = <C, J>: zero duality gap
    V2 = np.zeros((N, 64)); V2[np.arange(N), np.arange(N)] = 1.0   # X = I: feasible, not optimal
    y2, lmin2, lower2 = sdr.dual_certificate(C, V2)
    assert lmin2 < -1.0 and lower2 <= -N * N + 1e-9


# ------------------------------------------------------------------ cvxpy front end (duck-typed: cvxpy is not installed)
def test_cvxpy_adapter_extracts_the_reference_form():
    """get_qcqp_form's job (utilities.py:318-347) through qcqp_amd.cvxpy_adapter on a duck-typed cvxpy >= 1 problem:
    Boolean least squares as the README writes it (sum_squares(A x - b), square(x) == 1) plus a dense quadratic
    inequality and a >= constraint; one function per scalar constraint entry, symmetric P, maximise negated,
    non-quadratic expressions refused with the reference's messages, variable values restored."""
    from qcqp_amd.cvxpy_adapter import problem_from_cvxpy

    from duck_cvxpy import Var, Expr, Objective, Equality, Inequality, NonNeg, Prob

    rs = np.random.RandomState(0)
    n = 6
    A, b = rs.randn(9, n), rs.randn(9)
    G = rs.randn(n, n)
    g = rs.randn(n)
    x = Var((n,))
    x.value = np.arange(n, dtype=float)
    obj = Expr(lambda: np.sum((A.dot(x.value) - b) ** 2), ())
    cons = [Equality(Expr(lambda: x.value ** 2 - 1.0, (n,))),
            Inequality(Expr(lambda: x.value.dot(G).dot(x.value) + g.dot(x.value) - 3.0, ())),
            NonNeg(Expr(lambda: x.value[:2] + 2.0, (2,)))]
    p = problem_from_cvxpy(Prob(Objective('minimize', obj), cons, [x]))
    f = p.qcqp_form
    assert (f.n, f.m) == (n, n + 1 + 2) and p.objective.NAME == 'minimize'
    assert np.allclose(f.f0.P, A.T.dot(A)) and np.allclose(f.f0.qarray, -2 * A.T.dot(b)) and np.isclose(f.f0.r, b.dot(b))
    for i in range(n):
        E = np.zeros((n, n)); E[i, i] = 1.0
        assert np.allclose(np.asarray(f.fs[i].P), E) and f.fs[i].r == -1.0 and f.fs[i].relop == '=='
    assert np.allclose(f.fs[n].P, (G + G.T) / 2) and np.allclose(f.fs[n].qarray, g) and f.fs[n].relop == '<='
    # x_0 + 2 >= 0  ->  -x_0 - 2 <= 0
    assert f.fs[n + 1].relop == '<=' and f.fs[n + 1].qarray[0] == -1.0 and f.fs[n + 1].r == -2.0
    assert np.array_equal(x.value, np.arange(n, dtype=float))          # caller's values restored
    # maximise: negated like utilities.py:335-336
    pm = problem_from_cvxpy(Prob(Objective('maximize', obj), cons[:1], [x]))
    assert pm.objective.NAME == 'maximize' and np.allclose(pm.qcqp_form.f0.P, -A.T.dot(A))
    # variables adapter: (rows, cols) size, column-major values
    v = pm.variables()[0]
    assert v.size == (n, 1)
    v.value = np.ones((n, 1))
    assert x.value.shape == (n,) and np.all(x.value == 1.0)
    with pytest.raises(Exception, match='Objective is not quadratic'):
        problem_from_cvxpy(Prob(Objective('minimize', Expr(lambda: 0.0, (), quad=False)), [], [x]))
    with pytest.raises(Exception, match='Not all constraints are quadratic'):
        problem_from_cvxpy(Prob(Objective('minimize', obj), [Equality(Expr(lambda: x.value, (n,), quad=False))], [x]))


def test_cvxpy_adapter_keeps_bilinear_only_terms():
    """a bilinear term x_i x_j does not move f along either axis alone, so a pruning test on
    f(+-e_i) dropped P_ij silently.  Zero-diagonal objective x'Wx (MAXCUT-like), a constraint x0 x1 - 1 <= 0 and
    f = x0^2 + x0 x1 must all come back exactly; functions that do not share a variable are still not probed."""
    from qcqp_amd.cvxpy_adapter import problem_from_cvxpy
    from duck_cvxpy import Var, Expr, Objective, Equality, Inequality, Prob
    rs = np.random.RandomState(3)
    n = 7
    W = np.triu(rs.randn(n, n), 1)
    W = W + W.T                                           # zero diagonal, no linear term
    x = Var((n,))
    calls = [0]

    def objf():
        calls[0] += 1
        return x.value.dot(W).dot(x.value)
    cons = [Inequality(Expr(lambda: x.value[0] * x.value[1] - 1.0, ())),
            Inequality(Expr(lambda: x.value[0] ** 2 + x.value[0] * x.value[2] - 2.0, ())),
            Equality(Expr(lambda: x.value ** 2 - 1.0, (n,)))]
    f = problem_from_cvxpy(Prob(Objective('minimize', Expr(objf, ())), cons, [x])).qcqp_form
    assert np.allclose(np.asarray(f.f0.P), W, atol=1e-13) and np.allclose(f.f0.qarray, 0, atol=1e-13)
    P1 = np.zeros((n, n)); P1[0, 1] = P1[1, 0] = 0.5
    assert np.allclose(np.asarray(f.fs[0].P), P1, atol=1e-13) and f.fs[0].r == -1.0
    P2 = np.zeros((n, n)); P2[0, 0] = 1.0; P2[0, 2] = P2[2, 0] = 0.5
    assert np.allclose(np.asarray(f.fs[1].P), P2, atol=1e-13) and f.fs[1].r == -2.0
    # a problem whose functions share no pair of variables is extracted without any pair probe
    calls[0] = 0
    sep = [Equality(Expr(lambda: x.value ** 2 - 1.0, (n,)))]
    lin = Expr(lambda: (calls.__setitem__(0, calls[0] + 1), float(x.value[0]))[1], ())
    problem_from_cvxpy(Prob(Objective('minimize', lin), sep, [x]))
    assert calls[0] == 1 + 2 * n + 2 * (n + 1)


def test_dense_chain_geometry_tiles_the_constraints():
    """dense_chain_mw_kernel deals the constraints to (thread, slot) pairs and gives the objective to the serial thread
    (csrc/cd_dense_mw.h, mw_geometry): every constraint exactly once, the serial thread holds no constraint in slot 0, whole
    waves, at most 512 threads.  Host-only entry point (no device call)."""
    import ctypes
    from qcqp_amd import _ffi
    L = _ffi.lib()
    out = (ctypes.c_int * 4)()
    for m in list(range(0, 70)) + [127, 128, 129, 255, 256, 257, 447, 448, 449, 511, 512, 513, 1024, 2047, 3583, 3584, 3585, 5000]:
        assert L.qcqpmi_dense_chain_geometry(m, out) == 0
        SL, Tc, ts, T = out[0], out[1], out[2], out[3]
        assert T % 64 == 0 and Tc % 64 == 0 and 64 <= T <= 512 and Tc <= T and ts < T, (m, SL, Tc, ts, T)
        seen = set()
        for tid in range(Tc):
            for j in range(SL):
                k = 1 + tid + Tc * j
                if k <= m:
                    assert k not in seen
                    seen.add(k)
        assert seen == set(range(1, m + 1)), m
        # slot 0 of the serial thread is free for the objective
        assert ts >= Tc or 1 + ts > m, (m, ts, Tc)
        if SL == 1:
            assert ts == (m if m < Tc else Tc)


def test_own_lbfgs_matches_scipy_on_standard_problems():
    """The L-BFGS loop of the relaxation solver (qcqp_amd/sdr.py::_lbfgs) reaches the same minimisers as SciPy's L-BFGS-B with
    the same stopping rules: a convex quadratic (solution of A x = b) and the 20-dimensional Rosenbrock function."""
    from scipy.optimize import minimize
    from qcqp_amd.sdr import _lbfgs
    rs = np.random.RandomState(0)
    A = rs.randn(60, 60)
    A = A.dot(A.T) + np.eye(60)
    b = rs.randn(60)
    r = _lbfgs(lambda x: (0.5 * x.dot(A.dot(x)) - b.dot(x), A.dot(x) - b), np.zeros(60), 500, 1000, 1e-9, 1e-15)
    assert np.linalg.norm(A.dot(r.x) - b) < 1e-5 and r.nfev <= 1000

    def ros(x):
        f = np.sum(100 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
        g = np.zeros_like(x)
        g[:-1] = -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
        g[1:] += 200 * (x[1:] - x[:-1] ** 2)
        return f, g
    r = _lbfgs(ros, np.zeros(20), 2000, 4000, 1e-9, 1e-15)
    r2 = minimize(ros, np.zeros(20), jac=True, method='L-BFGS-B', options=dict(maxiter=2000, maxfun=4000, gtol=1e-9, ftol=1e-15, maxcor=20))
    assert r.fun < 1e-12 and np.allclose(r.x, 1.0, atol=1e-6) and np.allclose(r.x, r2.x, atol=1e-5)
    assert r.nfev < 2 * r2.nfev
    # limits are honoured
    r = _lbfgs(ros, np.zeros(20), 5, 4000, 1e-9, 1e-15)
    assert r.nit == 5


def test_reference_cd_is_chaotic_under_one_ulp(orc):
    """What parity statement a path with another summation order can make at all.  The reference's own coordinate descent with
    COUPLED constraints (qcqp.py:100-192), restated bit for bit by the oracle (golden G6 / G7: max |delta| = 0 against the
    reference), is run from x0 and from nextafter(x0) -- every entry ONE ULP up -- with the same keyed draws: a large share of
    the restarts ends more than the north star's 1e-6 away (phase 1 bisects the slack until the feasible set of a coordinate is
    nearly a point; its end points are roots of near-degenerate quadratics, d root / d coefficient = 1 / sqrt(discriminant)).
    The same experiment with /root/reference itself: profiles/r04_reference_sensitivity.md (41 % / 100 %).  Consequence: a
    free-running trajectory can be held to 1e-6 only by bit-identical arithmetic (the engine's reference-order mode, tested at
    1e-9); the default MFMA path is held to 1e-6 STEP BY STEP on the reference's own states
    (tests/test_gpu_scale.py::test_dense_default_path_follows_the_oracle_step_by_step).  The separable families of the headline
    (Boolean least squares: configs[0..2]) are not chaotic: there the free-running trajectories agree to 1e-9."""
    from qcqp_amd import problems

    def ends(prob, x0, r):
        rng = orc.Rng(orc.RNG_KEYED, 13)
        rng.set_restart(5 + r)
        return prob.improve_cd(x0, num_iters=5, rng=rng)[0]

    shares = {}
    for fam, funcs, R in (('dense', problems.dense_indefinite(100, 30, seed=11)[0], 16),
                          ('beam', problems.beamforming(50, 12, 4, seed=3)[0], 6),
                          ('bls', problems.boolean_least_squares(96, 40, seed=2)[0], 6)):
        prob = orc.Problem(funcs)
        X0 = 1.5 * np.random.RandomState(3).randn(prob.n, R)
        d = np.zeros(R)
        for r in range(R):
            a, b = ends(prob, X0[:, r], r), ends(prob, np.nextafter(X0[:, r], np.inf), r)
            d[r] = np.max(np.abs(a - b)) / (1 + np.max(np.abs(a)))
        shares[fam] = (float(np.mean(d > 1e-6)), float(np.median(d)))
    print('\nshare of restarts that one ulp of x0 moves by > 1e-6 (median move): %s' % shares)
    assert shares['dense'][0] >= 0.2
    assert shares['beam'][0] >= 0.8
    assert shares['bls'][0] == 0.0 and shares['bls'][1] < 1e-12


def test_global_best_of_populations_two_ranks():
    """dist.global_best_of_populations -- the one exchange of a streamed run (K populations): two ranks (threads here, the
    all-reduce is a barrier + sum) end with the same winners as the selection rule folded over both shards, population by
    population, incl. a tie on (bucket, objective) that the lower GLOBAL index must win and a rank whose restarts all failed."""
    import threading
    from qcqp_amd import dist
    K, n, world = 5, 7, 2
    rs = np.random.RandomState(4)
    f0 = rs.randn(world, K)
    mv = np.abs(rs.randn(world, K)) * 1e-5
    gi = np.stack([rs.randint(0, 100, K), 100 + rs.randint(0, 100, K)])
    f0[1, 2], mv[1, 2] = f0[0, 2], mv[0, 2]              # a tie: rank 0 holds the lower global index
    mv[0, 3] = 0.5                                        # another violation bucket: rank 1 wins whatever the objectives
    f0[1, 4], mv[1, 4] = np.inf, np.inf                   # every restart of rank 1 failed in population 4
    xs = rs.randn(world, K, n)
    slots, bar, res = [None, None], threading.Barrier(world), [None, None]

    def allreduce_for(rank):
        def allreduce(a):
            slots[rank] = np.array(a, dtype=np.float64)
            bar.wait()
            out = slots[0] + slots[1]
            bar.wait()
            return out
        return allreduce

    def run(rank):
        res[rank] = dist.global_best_of_populations(allreduce_for(rank), rank, world, f0[rank], mv[rank], gi[rank], xs[rank])
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    (k0, X0), (k1, X1) = res
    assert k0 == k1 and np.array_equal(X0, X1)
    for k in range(K):
        want = min(range(world), key=lambda g: dist.better_key(f0[g, k], mv[g, k], gi[g, k]))
        assert k0[k] == (int(gi[want, k]), float(f0[want, k]), float(mv[want, k])), k
        assert np.array_equal(X0[k], xs[want, k])
    assert k0[2][0] == gi[0, 2] and k0[3][0] == gi[1, 3] and k0[4][0] == gi[0, 4]
    # the same exchange with the keys through an ALL-GATHER (Engine.comm_allgather, round 5): same winners, and the bytes a rank
    # puts on the wire are 32 K (keys: bucket, f0, maxviol, global index -- integers as integers) + 8 K n (its table of points),
    # whatever the number of ranks -- not the (world, K, 3) table of round 4 through a sum-all-reduce
    sent = [dict(gather=0, reduce=0), dict(gather=0, reduce=0)]
    gslots, res2 = [None, None], [None, None]

    def allgather_for(rank):
        def allgather(a):
            a = np.ascontiguousarray(a)
            assert a.dtype == np.int64
            sent[rank]['gather'] += a.nbytes
            gslots[rank] = a.copy()
            bar.wait()
            out = np.stack([gslots[0], gslots[1]])
            bar.wait()
            return out
        return allgather

    def allreduce_counting(rank):
        inner = allreduce_for(rank)

        def allreduce(a):
            sent[rank]['reduce'] += np.asarray(a).nbytes
            return inner(a)
        return allreduce

    def run2(rank):
        big = gi[rank] + (1 << 40)                        # global indices beyond 2^32 travel exactly
        res2[rank] = dist.global_best_of_populations(allreduce_counting(rank), rank, world, f0[rank], mv[rank], big, xs[rank],
                                                     allgather=allgather_for(rank))
    th = [threading.Thread(target=run2, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    (g0, Y0), (g1, Y1) = res2
    assert g0 == g1 and np.array_equal(Y0, Y1) and np.array_equal(Y0, X0)
    assert [(i - (1 << 40), f, v) for i, f, v in g0] == k0
    for rank in range(world):
        assert sent[rank] == dict(gather=32 * K, reduce=8 * K * n), sent[rank]
    # one rank: no exchange at all
    ks, Xs = dist.global_best_of_populations(None, 0, 1, f0[0], mv[0], gi[0], xs[0])
    assert [k[0] for k in ks] == list(gi[0]) and np.array_equal(Xs, xs[0])


BENCH_WORKER = r'''
import json, os, sys, types
sys.path.insert(0, %(repo)r)
import numpy as np
from qcqp_amd import dist
import qcqp_amd.engine as engine_mod

BOOT = {}


class FakeEngine(object):
    """Stands in for qcqp_amd.engine.Engine on a box without a GPU: deterministic synthetic results per (rank, step, restart),
    the collectives over the job's file rendezvous -- everything bench.py does AROUND the kernels runs for real."""
    KERNEL_EVAL, KERNEL_CD1, KERNEL_CD2, KERNEL_SDR, KERNEL_ADMM = 0, 1, 2, 3, 4

    def __init__(self, form, device=0):
        self.n, self.device = form.n, device
        self.calls = []

    def comm_init(self, rank, world, uid):
        pass

    def comm_unique_id(self):
        return np.zeros(128, dtype=np.uint8)

    def cd_stream_reserve(self, K, R):
        pass

    def cd_stream_run(self, K, R, seed=0, seed_stride=1, first_index=0, first_stride=0, **kw):
        self.calls.append((K, R, seed, first_index))
        T = K * R
        rs = np.random.RandomState(seed %% 100000 + 7 * first_index)
        o = dict(sweeps1=np.ones(T, dtype=np.int64), visits2=np.full(T, 10 * self.n, dtype=np.int64), f0=rs.rand(T) + 5.0,
                 maxviol=np.zeros(T))
        o['best_index'] = np.array([int(np.argmin(o['f0'][p * R:(p + 1) * R])) for p in range(K)], dtype=np.int64)
        o['best_f0'] = np.array([o['f0'][p * R + o['best_index'][p]] for p in range(K)])
        o['best_maxviol'] = np.zeros(K)
        o['best_x'] = np.tile(o['best_f0'][:, None], (1, self.n))
        return o

    def kernel_ms(self, which):
        return 1.0

    def last_cd_kernel(self):
        return 'cd_life_kernel<3,band>'

    def sync(self):
        pass

    def comm_allreduce(self, values, op='max'):
        v = np.ascontiguousarray(np.atleast_1d(np.asarray(values, dtype=np.float64)))
        parts = BOOT['b'].allgather([float(a) for a in v])
        out = np.max(np.array(parts), axis=0) if op == 'max' else np.sum(np.array(parts), axis=0)
        v[...] = out
        return v

    def cd_life_version(self, version=0):
        pass

    def comm_allgather(self, arr):
        a = np.ascontiguousarray(arr)
        parts = BOOT['b'].allgather(a.ravel().tolist())
        return np.array(parts, dtype=a.dtype).reshape((len(parts),) + a.shape)

    def comm_barrier(self):
        BOOT['b'].barrier()


def fake_init_rccl(engine, rank, world, bootstrap=None):
    if 'b' not in BOOT:
        BOOT['b'] = dist.FileRendezvous(rank, world)
    return BOOT['b']


engine_mod.Engine = FakeEngine
dist.init_rccl = fake_init_rccl
sys.argv = [os.path.abspath(__file__), '--gpus', '2', '--steps', '3',      # (the self-spawned rank 1 re-runs THIS script)
             '--warmup', '1', '--n', '32', '--m-rows', '8', '--restarts', '16',
            '--no-secondary', '--no-cpu-baseline'] + %(extra)r
import bench
sys.exit(bench.main())
'''


@pytest.mark.parametrize('launch', ['self_spawn', 'launcher', 'self_spawn_file_comm'])
def test_bench_two_ranks_control_flow(tmp_path, launch):
    """bench.py --gpus 2 with a FAKE engine (no GPU here): the self-spawn of rank 1 -- or two processes started the way the
    driver's `python -m torch.distributed.run --nproc-per-node 2` starts them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment) --, the scheme decision taken identically on both ranks (one all-reduce), the streamed run per rank with the
    rank's own global restart indices, the exchange of a streamed run over the ranks (dist.global_best_of_populations through
    the engine's all-reduce), the max / sum reductions of the timing and work counters, ONE JSON line on rank 0 -- everything
    around the kernels runs for real; the collectives travel over the job's file rendezvous instead of RCCL."""
    import json
    import subprocess
    script = tmp_path / 'bench_worker.py'
    # (self_spawn_file_comm: `--comm file` -- qcqp_amd.dist.init_file_comm for real: the all-gather / all-reduce / barrier the engine
    #  methods are replaced with travel through the job's file rendezvous, as in the two-engines-one-GPU test of tests/test_gpu_life.py)
    script.write_text(BENCH_WORKER % dict(repo=REPO, extra=(['--comm', 'file', '--device', '0'] if launch == 'self_spawn_file_comm' else [])))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'QCQP_AMD_RDZV')}
    if launch.startswith('self_spawn'):
        pr = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=str(tmp_path))
        assert pr.returncode == 0, pr.stderr.decode()[-2000:]
        out = pr.stdout.decode()
    else:
        procs = []
        for r in range(2):
            e2 = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29%03d' % (os.getpid() % 1000),
                      TORCHELASTIC_RUN_ID='t%d' % os.getpid())
            procs.append(subprocess.Popen([sys.executable, str(script)], env=e2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=str(tmp_path)))
        outs = [p_.communicate(timeout=300) for p_ in procs]
        assert all(p_.returncode == 0 for p_ in procs), [o_[1].decode()[-1500:] for o_ in outs]
        assert not [l for l in outs[1][0].decode().splitlines() if l.startswith('{')]      # only rank 0 prints the line
        out = outs[0][0].decode()
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    assert out.rstrip('\n').splitlines()[-1] == lines[0]              # the line is the LAST thing on stdout
    assert len(lines[0]) < 4096                                       # compact: BENCH_r05's 20 KB line was not parsed by the driver

    def no_constants(name):
        raise AssertionError('non-strict JSON constant %s in the bench line' % name)
    d = json.loads(lines[0], parse_constant=no_constants)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'best'):
        assert key in d, key
    assert 'workload' in d['config'] and 'model' not in d['config']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'kernel_ms_per_launch'):
        assert key in d['roofline'], key
    assert 'secondary' not in d and 'schemes' not in d
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['config']['scheme'] == 'stream' and d['scaling'] == 'weak'
    # both ranks ran 3 steps of 16 restarts with 10 sweeps each: the whole-job value counts both
    assert abs(d['value'] * d['timed_region_s'] - 2 * 3 * 16 * 10) < 1e-6
    assert d['phase2_sweeps_per_restart'] == 10.0
    # the best point of the job: the smaller of the two ranks' winners, with its GLOBAL restart index (rank 1 owns 16..31)
    best = None
    for rank in range(2):
        rs = np.random.RandomState((2024 % 100000) + 7 * (16 * rank))
        f = rs.rand(3 * 16) + 5.0
        for p in range(3):
            i = int(np.argmin(f[p * 16:(p + 1) * 16]))
            cand = (float(f[p * 16 + i]), 16 * rank + i, p)
            best = cand if best is None or cand[0] < best[0] else best
    assert abs(d['best']['objective'] - best[0]) < 1e-12 and d['best']['global_restart_index'] == best[1] and d['best']['step'] == best[2]


def test_bench_line_is_compact_strict_json_and_last_on_stdout(tmp_path):
    """bench.headline_line / claim_stdout / emit_line: non-finite floats become null, a record that grew is cut back below the
    limit without losing a contract key, and whatever a C library printed through stdio (RCCL's banner at communicator creation,
    flushed at exit) cannot follow the line on the real stdout."""
    import json
    import subprocess
    sys.path.insert(0, REPO)
    import bench
    rec = {'metric': 'm', 'value': 1.0, 'unit': 'u', 'n_gpus': 1, 'steps': 2, 'warmup': 1, 'ms_per_step': float('nan'),
           'config': {'workload': 'w' * 300, 'note': 'x' * 3000}, 'roofline': {'frac': np.float64(0.5), 'timing': 't' * 2000, 'traffic': float('inf')},
           'cpu_baseline': {'value': np.float32(2.0), 'sample': 's' * 1500}, 'best': {'objective': 1.0, 'index': np.int64(3)},
           'phase1': {'note': 'p' * 500}}
    line = bench.headline_line(rec)
    assert len(line) < bench.HEADLINE_LIMIT and '\n' not in line
    d = json.loads(line, parse_constant=lambda name: (_ for _ in ()).throw(AssertionError(name)))
    assert d['ms_per_step'] is None and d['roofline']['traffic'] is None and d['roofline']['frac'] == 0.5 and d['best']['index'] == 3
    assert all(k in d for k in ('metric', 'value', 'unit', 'config', 'roofline', 'cpu_baseline', 'best'))
    prog = tmp_path / 'emit.py'
    prog.write_text('''
import ctypes, os, sys
sys.path.insert(0, %r)
import bench
fd = bench.claim_stdout()
libc = ctypes.CDLL(None)
libc.puts(b"RCCL version : banner through C stdio")      # buffered by stdio until exit
print("python chatter")
bench.emit_line(fd, '{"value": 1}')
''' % REPO)
    pr = subprocess.run([sys.executable, str(prog)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert pr.returncode == 0, pr.stderr.decode()[-1500:]
    assert pr.stdout.decode() == '{"value": 1}\n'
    assert b'banner' in pr.stderr and b'chatter' in pr.stderr


def test_objective_factor_found_verified_or_refused():
    """qcqp_amd.lowrank.objective_factor (pivoted Cholesky, NumPy only): a least-squares objective gives a factor of rank rows(A) that
    reproduces P0 entry by entry; an indefinite objective (MAXCUT), a rank above the limit, a matrix that is not PSD by a hair and a
    non-square input are refused (None) -- the caller then multiplies with P0 as before."""
    from qcqp_amd import lowrank, problems
    funcs = problems.boolean_least_squares(96, 24, seed=3)[0]
    P0 = funcs[0][0]
    P0 = P0.toarray() if hasattr(P0, 'toarray') else np.asarray(P0)
    L = lowrank.objective_factor(P0, max_rank=48)
    assert L is not None and L.shape == (96, 24)
    assert np.max(np.abs(L @ L.T - P0)) <= 1e-12 * np.max(np.abs(P0))
    assert lowrank.objective_factor(P0, max_rank=23) is None                       # rank above the limit
    assert lowrank.objective_factor(P0 - 1e-6 * np.eye(96), max_rank=96) is None   # indefinite by a hair (the factor would not verify)
    Pm = problems.maxcut(40, 0.5, seed=1)[0][0][0]
    Pm = Pm.toarray() if hasattr(Pm, 'toarray') else np.asarray(Pm)
    assert lowrank.objective_factor(Pm) is None                                    # zero diagonal, indefinite
    assert lowrank.objective_factor(np.zeros((8, 8))) is None and lowrank.objective_factor(np.ones((3, 4))) is None
    full = np.eye(16) * 2.0
    Lf = lowrank.objective_factor(full, max_rank=16)
    assert Lf is not None and np.allclose(Lf @ Lf.T, full, atol=1e-14)


def test_build_units_cover_every_source_file():
    """qcqp_amd/_build.py derives what a translation unit is built from by scanning its #include lines (round 4 kept the lists by
    hand and missed a header: an edit to cd_phase1_sep.h rebuilt nothing, and a GPU box could run yesterday's phase 1).  Every
    file under csrc/ belongs to a unit, every quoted include of every file resolves, and a header newer than an object makes
    that unit stale."""
    import re
    from qcqp_amd import _build
    files = sorted(f for f in os.listdir(_build.SRC) if f.endswith(('.hip', '.h', '.inc')))
    covered = set(_build.SOURCES)
    assert not [f for f in files if f not in covered], [f for f in files if f not in covered]
    inc = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)
    for f in files:
        for name in inc.findall(open(os.path.join(_build.SRC, f)).read()):
            assert os.path.exists(os.path.normpath(os.path.join(_build.SRC, name))), (f, name)
    for unit in _build.TRANSLATION_UNITS:
        deps = _build.unit_sources(unit)
        assert os.path.join(_build.SRC, unit) in deps
    assert os.path.join(_build.SRC, 'cd_phase1_sep.h') in _build.unit_sources('cd_queue.hip')
    assert os.path.join(_build.SRC, 'cd_phase1_sep.h') in _build.unit_sources('capi.hip')
    assert os.path.join(_build.SRC, 'cd_phase1_sep.h') in _build.unit_sources('cd_life.hip')
    # staleness: an object older than one of its sources is rebuilt
    obj = _build._obj('cd_life.hip')
    if os.path.exists(obj):
        hdr = os.path.join(_build.SRC, 'cd_phase1_sep.h')
        st = os.stat(hdr)
        try:
            os.utime(hdr, (st.st_atime, os.path.getmtime(obj) + 10))
            assert _build._stale(obj, _build._unit_deps('cd_life.hip'))
        finally:
            os.utime(hdr, (st.st_atime, st.st_mtime))


def test_host_thread_budget_follows_the_cgroup_quota_and_the_ranks(monkeypatch):
    """qcqp_amd/_threads.py (round 6: a 256-thread BLAS pool in a container with 16 cores of quota got the host thread that waits for
    the GPU throttled -- profiles/r06_timed_region.md): usable cores = min(affinity, quota), the BLAS pools get half of a rank's share
    (at most 8, at least 1), the environment is only filled where the user has not spoken."""
    import os
    from qcqp_amd import _threads
    cores = _threads.usable_cores()
    assert 1 <= cores <= (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            assert cores <= max(1, int(quota) // int(period))
    except OSError:
        pass
    for v in ('LOCAL_WORLD_SIZE', 'WORLD_SIZE', 'QCQP_LOCAL_RANKS'):
        monkeypatch.delenv(v, raising=False)
    one = _threads.blas_threads()
    assert one == max(1, min(8, cores // 2))
    monkeypatch.setenv('WORLD_SIZE', '8')
    assert _threads.local_ranks() == 8 and _threads.blas_threads() == max(1, min(8, cores // 16))
    monkeypatch.setenv('OPENBLAS_NUM_THREADS', '3')
    monkeypatch.delenv('OMP_NUM_THREADS', raising=False)
    _threads.set_blas_env()
    assert os.environ['OPENBLAS_NUM_THREADS'] == '3' and os.environ['OMP_NUM_THREADS'] == str(_threads.blas_threads())
    with _threads.blas_limit():         # usable with or without threadpoolctl
        import numpy as np
        assert float(np.ones((8, 8)).dot(np.ones(8)).sum()) == 64.0


def test_multi_class_families_are_what_the_kinds_take_and_the_oracle_solves_them():
    """problems.multi_class (round 6: the families of cd_life_kernel's GENK / LINK kinds): every coordinate is constrained, at most two
    constraints per coordinate, at most four distinct constraint lists; the ORACLE's improve_coord_descent (qcqp.py:181-192) drives a
    random start of each family to a point that is feasible within the phase-1 slack (viol_tol = 1e-2)."""
    import numpy as np
    from oracle import oracle as orc
    from qcqp_amd import problems
    from qcqp_amd.form import QCQPForm
    n = 24
    for fam in ('box3', 'ann2', 'lin2', 'cut2'):
        funcs = problems.multi_class(fam, n)
        form = QCQPForm.from_arrays(funcs)
        lists = {}
        for (P, q, r, relop) in funcs[1:]:
            P = P.toarray() if hasattr(P, 'toarray') else np.asarray(P)
            nz = np.flatnonzero(np.abs(P).sum(axis=0) + np.abs(np.asarray(q)))
            assert nz.size == 1 and np.count_nonzero(P - np.diag(np.diag(P))) == 0, fam      # separable: one coordinate per constraint
            i = int(nz[0])
            lists.setdefault(i, []).append((float(P[i, i]), float(np.asarray(q)[i]), float(r), relop))
        assert sorted(lists) == list(range(n)), fam                                      # every coordinate constrained
        assert max(len(v) for v in lists.values()) <= 2
        assert len({tuple(v) for v in lists.values()}) <= 4
        assert form.m == sum(len(v) for v in lists.values())
        prob = orc.Problem(funcs)
        rng = orc.Rng(orc.RNG_KEYED, 7)
        rng.set_restart(3)
        x0 = orc.keyed_normal_matrix(7, n, 1, first_index=3)[:, 0]
        x, s1, s2 = prob.improve_cd(x0, num_iters=200, rng=rng)
        assert prob.max_violation(x) < 1e-2, (fam, prob.max_violation(x))
