"""Builds libqcqp_mi.so (hand-written HIP for gfx950) in-tree with hipcc.

Several translation units, compiled in parallel and only when one of their sources changed (the big one takes two
minutes): objects under qcqp_amd/_obj/ (git-ignored), linked into qcqp_amd/libqcqp_mi.so."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, '_obj')
LIB = os.path.join(HERE, 'libqcqp_mi.so')
HEADER = os.path.join(REPO, 'include', 'qcqp_mi.h')

# translation unit -> everything it includes (rebuilt when any of these is newer than its object)
UNITS = {
    'capi.hip': ['capi.hip', 'capi_admm.inc', 'capi_units.inc', 'capi_dense.inc', 'kernels.hip', 'kernels.h', 'onevar.h', 'philox.h',
                 'cd_phase2.h', 'cd_phase2_rs.h', 'cd_phase2_q.h', 'admm.h', 'admm_fused.h', 'cd_queue.h', 'gemm_pk.h', 'cd_general.h', 'cd_dense.h', 'cd_dense_mw.h',
                 'sdr_solve.h'],
    'admm_fused.hip': ['admm_fused.hip', 'admm_fused.h', 'onevar.h', 'philox.h'],
    'cd_queue.hip': ['cd_queue.hip', 'cd_queue.h', 'cd_phase2_q.h', 'cd_phase2_rs.h', 'cd_phase2.h', 'kernels.h', 'onevar.h', 'philox.h'],
}
SOURCES = sorted(set(sum(UNITS.values(), [])))


def _obj(unit):
    return os.path.join(OBJ, os.path.splitext(unit)[0] + '.o')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _unit_deps(unit):
    return [os.path.join(SRC, s) for s in UNITS[unit]] + [HEADER]


def needs_build():
    return any(_stale(_obj(u), _unit_deps(u)) for u in UNITS) or _stale(LIB, [_obj(u) for u in UNITS if os.path.exists(_obj(u))])


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).  -ffp-contract=off keeps the
    scalar decision logic in the same unfused IEEE arithmetic as the reference's NumPy code."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(OBJ, exist_ok=True)
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-I' + os.path.join(REPO, 'include')]
    if os.environ.get('QCQPMI_DN_PROFILE') == '1':     # stage timers of the dense chain kernels (tools/dense_stage_profile.py)
        flags.append('-DDN_PROFILE=1')

    def compile_unit(unit):
        cmd = [hipcc] + flags + ['-c', os.path.join(SRC, unit), '-o', _obj(unit)]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)

    todo = [u for u in UNITS if force or _stale(_obj(u), _unit_deps(u))]
    with ThreadPoolExecutor(max_workers=max(1, len(todo))) as ex:
        list(ex.map(compile_unit, todo))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + [_obj(u) for u in UNITS] + ['-ldl']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    import sys
    build(force='--force' in sys.argv, verbose=True)
