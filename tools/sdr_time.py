import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm
funcs, _, _ = problems.boolean_least_squares(8, 8, seed=1)
e = Engine(QCQPForm.from_arrays(funcs))
for N in (257, 513, 1025, 2049, 4096):
    rs = np.random.RandomState(0)
    C = rs.randn(N, N); C = (C + C.T) / 2
    e.sdr_solve_unitdiag(C, max_sweeps=1, tol=0.0)
    t0 = time.time(); V, hist, sw = e.sdr_solve_unitdiag(C, max_sweeps=8, tol=0.0); t1 = time.time()
    print('N=%5d: %d sweeps + 2 passes in %.3f s -> %.2f us per coordinate step' % (N, sw, t1 - t0, (t1 - t0) / ((sw + 2) * N) * 1e6))
