#!/usr/bin/env python
"""How sensitive is the REFERENCE's own coordinate descent to its input?  (build container only)

Runs /root/reference's improve_coord_descent (qcqp.py:181-192) twice per restart -- from x0 and from x0 with every entry
moved by ONE ULP -- with the same np.random seed, on the coupled-constraint families of tests/test_gpu_scale.py
(dense indefinite n = 100, m = 30; beamforming n = 100), and prints how far apart the two runs end.  The answer decides
what parity statement a path with another summation order (MFMA products) can make at all: if one ulp of input moves the
reference's own result by more than the north star's 1e-6, no free-running trajectory of any other arithmetic can be held
to 1e-6 -- only bit-identical arithmetic (qcqpmi_cd_reference_order) or a step-by-step comparison on the reference's own
states (qcqpmi_cd_dense_block_step) can.  The same experiment through the oracle runs in tests/test_host_cpu.py.

Usage: python tools/reference_sensitivity.py [restarts] > profiles/r04_reference_sensitivity.md"""
import os
import sys
import tempfile
import types
import warnings

import numpy as np
import scipy.sparse as sp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
if not os.path.isdir(REF):
    sys.exit('reference not present: this experiment only runs in the build container')
sys.path.insert(0, REPO)
from qcqp_amd import problems  # noqa: E402

m = types.ModuleType('cvxpy')
m.__path__ = []
u = types.ModuleType('cvxpy.utilities')
u.QuadCoeffExtractor = object
lo = types.ModuleType('cvxpy.lin_ops')
lo.__path__ = []
lu = types.ModuleType('cvxpy.lin_ops.lin_utils')
sys.modules.update({'cvxpy': m, 'cvxpy.utilities': u, 'cvxpy.lin_ops': lo, 'cvxpy.lin_ops.lin_utils': lu})
sys.path.insert(0, REF)
os.chdir(tempfile.mkdtemp(prefix='qcqp_sens_'))
import qcqp.utilities as U  # noqa: E402
import qcqp.qcqp as Q  # noqa: E402

warnings.simplefilter('ignore')


def ref_prob(funcs):
    fs = []
    for (P, q, r, relop) in funcs:
        n = np.asarray(q).size
        fs.append(U.QuadraticFunction(sp.csr_matrix(P), sp.csc_matrix(np.asarray(q).reshape(n, 1)), r, relop))
    return U.QCQPForm(fs[0], fs[1:])


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    iters = 5
    print('# Sensitivity of the reference\'s own coordinate descent to one ulp of input (tools/reference_sensitivity.py)\n')
    print('`improve_coord_descent(x0, prob, num_iters=%d)` of /root/reference, run from x0 and from `nextafter(x0, inf)` '
          '(every entry one ulp up) with the same `np.random.seed`; d = max |x - x\'| / (1 + max |x|) of the two results.\n' % iters)
    print('| family | restarts | median d | 90 % | max | share with d > 1e-6 | share with d > 1e-9 |')
    print('|---|---|---|---|---|---|---|')
    for name, funcs in (('dense indefinite n=100 m=30 (seed 11)', problems.dense_indefinite(100, 30, seed=11)[0]),
                        ('beamforming n=100 m=16 (seed 3)', problems.beamforming(50, 12, 4, seed=3)[0])):
        prob = ref_prob(funcs)
        n = prob.n
        X0 = 1.5 * np.random.RandomState(3).randn(n, R)
        d = np.zeros(R)
        for r in range(R):
            np.random.seed(100 + r)
            a = Q.improve_coord_descent(X0[:, r].copy(), prob, num_iters=iters)
            np.random.seed(100 + r)
            b = Q.improve_coord_descent(np.nextafter(X0[:, r], np.inf), prob, num_iters=iters)
            d[r] = np.max(np.abs(a - b)) / (1 + np.max(np.abs(a)))
            sys.stderr.write('%s restart %d: %.2e\n' % (name, r, d[r]))
        print('| %s | %d | %.1e | %.1e | %.1e | %.2f | %.2f |' % (name, R, np.median(d), np.percentile(d, 90), d.max(),
                                                                 np.mean(d > 1e-6), np.mean(d > 1e-9)))


if __name__ == '__main__':
    main()
