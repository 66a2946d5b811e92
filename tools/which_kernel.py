"""Prints the phase-2 kernel the library dispatches to for the configurations the parity tests use (tests assert these names)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from conftest import funcs_from_npz, load_golden
from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm

def run(tag, funcs, R=8, generic=False):
    e = Engine(QCQPForm.from_arrays(funcs))
    e.L.qcqpmi_debug_profile(e.h, 2 if generic else 0, None)
    n = e.n
    e.upload(np.sign(np.random.RandomState(0).randn(n, R)) * 1.00001)
    e.cd_run(phase1=True, num_iters=3, seed=1)
    print('%-28s generic=%d n=%d -> %s' % (tag, generic, n, e.last_cd_kernel()))

for g in (False, True):
    for name in ['bls10', 'bls32', 'bls64', 'maxcut12']:
        run('golden ' + name, funcs_from_npz(load_golden('g6_cd_' + name)), generic=g)
    for nm, n, mr in [('bls', 96, 40), ('bls', 250, 100), ('maxcut', 130, 0), ('bls', 48, 20), ('bls', 1024, 256), ('bls', 1040, 256), ('bls', 2048, 64)]:
        funcs = problems.boolean_least_squares(n, mr, seed=2)[0] if nm == 'bls' else problems.maxcut(n, 0.5, seed=3, weighted=True)[0]
        run('%s n=%d' % (nm, n), funcs, generic=g)
