import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from qcqp_amd import dist, problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm
funcs, _, _ = problems.boolean_least_squares(1024, 256, seed=1)
eng = Engine(QCQPForm.from_arrays(funcs))
dist.init_rccl(eng, 0, 1)
R = 4096
T = np.zeros(4)
for k in range(60):
    eng.sync(); t0 = time.perf_counter()
    eng.randn(R, seed=100 + k, first_index=0); t1 = time.perf_counter()
    out = eng.cd_run(phase1=True, seed=100 + k, first_index=0); t2 = time.perf_counter()
    best = eng.comm_select_best(1e-4, index_offset=0); t3 = time.perf_counter()
    if k >= 10:
        T += [t1 - t0, t2 - t1, t3 - t2, eng.kernel_ms(Engine.KERNEL_CD2) / 1e3 + eng.kernel_ms(Engine.KERNEL_CD1) / 1e3]
T /= 50
print('per step: randn call %.3f ms, cd_run %.3f ms (phase-1 + phase-2 kernels %.3f ms), comm_select_best %.3f ms' % (T[0] * 1e3, T[1] * 1e3, T[3] * 1e3, T[2] * 1e3))
