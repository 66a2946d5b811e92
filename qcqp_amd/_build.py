"""Builds libqcqp_mi.so (hand-written HIP for gfx950) in-tree with hipcc.

Several translation units, compiled in parallel and only when one of their sources changed (the big one takes two
minutes): objects under qcqp_amd/_obj/ (git-ignored), linked into qcqp_amd/libqcqp_mi.so."""
import os
import re
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, '_obj')
LIB = os.path.join(HERE, 'libqcqp_mi.so')
HEADER = os.path.join(REPO, 'include', 'qcqp_mi.h')

# translation units; what each one includes is found by scanning its `#include "..."` lines recursively (round 4 shipped a
# hand-kept list that missed a header: an edit to it rebuilt nothing)
TRANSLATION_UNITS = ['capi.hip', 'admm_fused.hip', 'cd_queue.hip', 'cd_life.hip']
_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _scan(path, seen):
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return
    seen.add(path)
    with open(path) as f:
        text = f.read()
    for inc in _INC.findall(text):
        _scan(os.path.join(os.path.dirname(path), inc), seen)


def unit_sources(unit):
    """Absolute paths of every file the translation unit is built from (itself, headers, .inc files, the C ABI header)."""
    seen = set()
    _scan(os.path.join(SRC, unit), seen)
    return sorted(seen)


UNITS = {u: [os.path.relpath(p, SRC) for p in unit_sources(u)] for u in TRANSLATION_UNITS}
SOURCES = sorted(set(sum(UNITS.values(), [])))


def _obj(unit):
    return os.path.join(OBJ, os.path.splitext(unit)[0] + '.o')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _unit_deps(unit):
    return unit_sources(unit) + [HEADER]


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
