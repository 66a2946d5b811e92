"""CPU-only tests of the host side: C-ABI symbol export, problem containers, sharding and the
selection rule (incl. a 2-process gloo run).  No GPU compute is called here."""
import ctypes
import os
import re
import subprocess
import sys

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
    assert L.qcqpmi_abi_version() == 1
    assert L.qcqpmi_device_count() >= 0


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
rank, local, world = dist.env_world()
boot = dist.GlooBootstrap()
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
gkey, gx = dist.global_best_gloo(boot, key, out[:, key[2] - first])
np.save(os.path.join(%(tmp)r, 'rank%%d.npy' %% rank), np.concatenate([[gkey[0], gkey[1], gkey[2]], gx]))
boot.barrier()
'''


def test_two_process_gloo_selection(tmp_path, orc):
    """world_size 2 on CPU: restarts sharded by global index, the final exchange picks the same
    global best on both ranks, identical to the single-process answer."""
    from qcqp_amd import dist, problems
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(repo=REPO, tmp=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', '29577', str(script)]
    subprocess.run(cmd, check=True, env=env, timeout=600, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    r0 = np.load(tmp_path / 'rank0.npy')
    r1 = np.load(tmp_path / 'rank1.npy')
    assert np.array_equal(r0, r1)
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
    assert np.array_equal(r0[3:], xs[:, key[2]])
