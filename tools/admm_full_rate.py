"""improve(ADMM) in the FULL eigenbasis (constraints of any rank; the multi-launch path): rate and, under rocprofv3
--kernel-trace --stats, the split over the kernels.  usage: admm_full_rate.py [antennas=256] [R=256] [iters=30]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm

na = int(sys.argv[1]) if len(sys.argv) > 1 else 256
R = int(sys.argv[2]) if len(sys.argv) > 2 else 256
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
funcs, _, _ = problems.beamforming(na, 8, 32, seed=1)
form = QCQPForm.from_arrays(funcs)
e = Engine(form)
lm = np.zeros((form.m, form.n)); Q = np.zeros((form.m, form.n, form.n))
for k, f in enumerate(form.fs):
    lm[k], Q[k] = np.linalg.eigh(np.asarray(f.P))
e.admm_set_eig(lm, Q)
for rep in range(3):
    e.randn(R, seed=3)
    e.sync()
    t0 = time.perf_counter()
    o = e.admm_run(1.0, None, phase1=True, num_iters=iters)
    e.sync()
    dt = time.perf_counter() - t0
    its = float(o['iters1'].sum() + o['iters2'].sum())
    fl = 4.0 * form.n * form.n * form.m
    print('n=%d m=%d R=%d: %.1f ms, %.3e restart-iterations/s, %.2f TFLOP/s (4 n^2 m per restart-iteration), %.3f ms per iteration of the population; kernel %s'
          % (form.n, form.m, R, 1e3 * dt, its / dt, its * fl / dt / 1e12, 1e3 * dt / max(1, int(o['iters1'].max() + o['iters2'].max())), e.last_admm_kernel()[0]))
