"""Digest of the dense-constraint coordinate descent on fixed seeds (configs[4] family generated on the device): run before
and after a change that must not move a bit.  usage: python tools/dense_hash.py"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems
from qcqp_amd.engine import Engine

for (n, m, R, iters) in ((256, 64, 512, 3), (1024, 256, 512, 2), (512, 100, 2048, 2), (250, 33, 100, 3)):
    form = problems.dense_indefinite_generated(n, m, seed=7)
    e = Engine(form)
    e.dense_chain_mode(1 if '--one-wave' in sys.argv else 0)
    e.randn(R, seed=6)
    e.sync()
    t0 = time.perf_counter()
    out = e.cd_run(phase1=True, num_iters=iters, seed=6)
    dt = time.perf_counter() - t0
    X = e.download()
    h = hashlib.sha256()
    for a in (X, out['f0'], out['maxviol'], out['sweeps1'], out['visits2'], out['accepted2']):
        h.update(np.ascontiguousarray(a).tobytes())
    print('n %4d m %3d R %4d iters %d: %s  kernel %s  %.3f s  best f0 %.12g' % (n, m, R, iters, h.hexdigest()[:24], e.last_cd_kernel(), dt, float(np.min(out['f0']))))
