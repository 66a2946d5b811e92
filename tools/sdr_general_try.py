import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from qcqp_amd import problems, sdr
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm
# (1) cross-check with the mixing solver on a Boolean problem pushed through the dense path
n = 12
funcs, _, _ = problems.boolean_least_squares(n, 16, seed=1)
form = QCQPForm.from_arrays(funcs)
e0 = Engine(form)
X0, b0, i0 = sdr.solve_sdr(e0, form)
# general solver needs the dense path: add a tiny off-diagonal coupling-free trick -> use a rotated copy: make constraints dense by
# expressing x_i^2 = 1 through dense matrices is not possible; instead test on a dense family below
print('mixing bound', b0)
# (2) dense indefinite family
funcs, _, _ = problems.dense_indefinite(24, 6, seed=11)
form = QCQPForm.from_arrays(funcs)
e = Engine(form)
e.L.qcqpmi_debug_profile(e.h, 32 << 4, None)
t0 = time.time(); X, bound, info = sdr.solve_sdr_general(e, form, verbose=True); t1 = time.time()
lmin, S = sdr.dual_certificate_general(form, info['y'], info['yN'])
print('bound %.8g dual value %.8g lambda_min(S) %.3e evals %d time %.1f s rank %d' % (bound, info['dual_value'], lmin, info['evals'], t1 - t0, info['rank']))
print('eig X min %.2e  X_nn %.6f' % (np.linalg.eigvalsh(X)[0], X[-1, -1]))
