"""Builds libqcqp_mi.so (hand-written HIP for gfx950) in-tree with hipcc."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libqcqp_mi.so')
SOURCES = ['capi.hip', 'capi_admm.inc', 'capi_units.inc', 'kernels.hip', 'kernels.h', 'onevar.h', 'philox.h', 'cd_phase2.h',
           'cd_phase2_rs.h', 'cd_phase2_q.h', 'admm.h', 'gemm_pk.h', 'cd_general.h', 'cd_dense.h', 'capi_dense.inc', 'sdr_solve.h']


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(SRC, s) for s in SOURCES] + [os.path.join(REPO, 'include', 'qcqp_mi.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).  -ffp-contract=off keeps the
    scalar decision logic in the same unfused IEEE arithmetic as the reference's NumPy code."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-ffp-contract=off', '-I' + os.path.join(REPO, 'include'),
           '-o', LIB, os.path.join(SRC, 'capi.hip'), '-ldl']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force=True, verbose=True)
