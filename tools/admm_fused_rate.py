"""Throughput of improve(ADMM) at BASELINE.json configs[3] size (n=1024, m=80, rho=1): fused persistent kernel vs the
multi-launch path.  usage: admm_fused_rate.py [R=1024] [iters=200]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import lowrank, problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
funcs, _, _ = problems.beamforming(512, 16, 64, seed=1)
form = QCQPForm.from_arrays(funcs)
e = Engine(form)
lam, Bv, qhat, info = lowrank.reduced_bases(e, form)
e.admm_set_basis(lam, Bv, qhat)
print('n=%d m=%d rp=%d R=%d num_iters=%d' % (form.n, form.m, info['rp'], R, iters))
import ctypes as C
prof = '--prof' in sys.argv
for fused in (True, False, 2):
    e.admm_fused(fused)
    e.L.qcqpmi_debug_profile(e.h, 1 if (prof and fused) else 0, None)
    for rep in range(2):
        e.randn(R, seed=5)
        e.sync()
        t0 = time.time()
        out = e.admm_run(1.0, None, phase1=True, num_iters=iters)
        dt = time.time() - t0
    its = float(out['iters1'].sum() + out['iters2'].sum())
    name, cw = e.last_admm_kernel()
    kms = e.kernel_ms(4)
    print('%-18s C=%2d: %.4f s wall, %.3e restart-iterations -> %.3e /s (wall); iterations of the longest restart %d + %d; feasible %d; '
          'timer[4] %.3f ms' % (name, cw, dt, its, its / dt, out['iters1'].max(), out['iters2'].max(), int((out['maxviol'] < 1e-2).sum()), kms))
    if prof and fused:
        pr = np.zeros(16, dtype=np.int64)
        e.L.qcqpmi_debug_admm_profile(e.h, pr.ctypes.data_as(C.POINTER(C.c_int64)))
        nit = max(1, int(pr[9]))
        names = ['z-update', 'partial product', 'exchange 1', 'sums', 'secular', 'exchange 2', 'gather', 'book', 'loop top']
        tot = float(pr[:9].sum())
        print('   per iteration of tile 0 (s_memtime cycles), %d iterations, total %.0f cycles (launch / longest chain: %.2f us per iteration): ' % (nit, tot / nit, kms * 1e3 / max(1, int(out['iters1'].max() + out['iters2'].max())))
              + ', '.join('%s %.0f (%.0f%%)' % (nm, pr[k] / nit, 100 * pr[k] / tot) for k, nm in enumerate(names)))
        print('   wave 0 alone: fragment stream of the z-update %.0f, z-update until its element-wise part is done %.0f, fragment stream of the partial product %.0f cycles'
              % (pr[10] / nit, pr[12] / nit, pr[11] / nit))
