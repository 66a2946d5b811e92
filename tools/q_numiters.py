import sys
sys.path.insert(0,'/root/repo')
import numpy as np
from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm
for n in (48, 256, 1024):
    funcs,_,_ = problems.boolean_least_squares(n, n//4, seed=3)
    e = Engine(QCQPForm.from_arrays(funcs))
    for ni in (0, 1, 2, 3, 5):
        res = []
        for mode in (64 << 4, 0):
            e.L.qcqpmi_debug_profile(e.h, mode, None)
            e.randn(100, seed=9)
            out = e.cd_run(num_iters=ni, seed=9)
            res.append((e.download(), out['visits2'].copy(), out['f0'].copy(), e.last_cd_kernel()))
        d = np.max(np.abs(res[0][0]-res[1][0]))
        print('n', n, 'num_iters', ni, res[0][3], res[1][3], 'max|dx|', d, 'visits equal', np.array_equal(res[0][1], res[1][1]), 'f0 rel', np.max(np.abs(res[0][2]-res[1][2])/(1+np.abs(res[0][2]))))
