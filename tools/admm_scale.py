"""BASELINE.json configs[3] (secondary-user beamforming, n = 2 x 512 real variables, 16 SINR + 64 interference constraints,
improve(ADMM, rho=1)): setup time and restart-iterations / s of the reduced-basis path and of the full-eigenbasis path,
and their agreement on a few restarts.  usage: python tools/admm_scale.py [nant] [R] [iters] [full]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems, lowrank
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm

nant = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 40
full = len(sys.argv) > 4 and sys.argv[4] == 'full'
funcs, _, _ = problems.beamforming(nant, 16, 64, seed=1)
form = QCQPForm.from_arrays(funcs)
n, m = form.n, form.m
t0 = time.time(); e = Engine(form); t_up = time.time() - t0
t0 = time.time()
lam, Bv, qhat, info = lowrank.reduced_bases(e, form)
e.admm_set_basis(lam, Bv, qhat)
t_setup = time.time() - t0
print('n %d m %d: engine build %.2f s, reduced-basis setup %.2f s (rank %d, rp %d)' % (n, m, t_up, t_setup, info['rank'].max(), info['rp']))
rho = 1.0
for rr in (R,):
    e.randn(rr, seed=3)
    X0 = e.download()
    e.admm_run(rho, None, phase1=True, num_iters=2)       # warm
    e.upload(X0)
    e.sync(); t0 = time.time()
    out = e.admm_run(rho, None, phase1=True, num_iters=iters)
    e.sync(); dt = time.time() - t0
    its = float(out['iters1'].sum() + out['iters2'].sum())
    print('reduced basis: R %d, %d iterations each phase: %.3f s, %.3e restart-iterations/s, feasible %d, f0 median %.4f, secular kernel %.3f ms'
          % (rr, iters, dt, its / dt, int((out['maxviol'] < 1e-2).sum()), np.median(out['f0']), e.kernel_ms(4)))
    Xl = e.download(); ol = out
if full:
    t0 = time.time()
    lm = np.zeros((m, n)); Q = np.zeros((m, n, n))
    for k, f in enumerate(form.fs):
        lm[k], Q[k] = np.linalg.eigh(np.asarray(f.P))
    t_eig = time.time() - t0
    e2 = Engine(form)
    t0 = time.time(); e2.admm_set_eig(lm, Q); t_set = time.time() - t0
    Rf = min(R, 128)
    e2.upload(X0[:, :Rf])
    e2.admm_run(rho, None, phase1=True, num_iters=2)
    e2.upload(X0[:, :Rf])
    e2.sync(); t0 = time.time()
    out2 = e2.admm_run(rho, None, phase1=True, num_iters=iters)
    e2.sync(); dt = time.time() - t0
    its = float(out2['iters1'].sum() + out2['iters2'].sum())
    print('full eigenbasis: host eigh %.1f s, upload+pack %.1f s; R %d: %.3f s, %.3e restart-iterations/s, secular kernel %.3f ms' % (t_eig, t_set, Rf, dt, its / dt, e2.kernel_ms(4)))
    Xf = e2.download()
    d = np.max(np.abs(Xf - Xl[:, :Rf]), axis=0)
    print('reduced vs full basis: max |dx| per restart: median %.2e max %.2e; f0 rel diff max %.2e; iters equal %d of %d' % (
        np.median(d), d.max(), np.max(np.abs(out2['f0'] - ol['f0'][:Rf]) / (1 + np.abs(out2['f0']))),
        int((out2['iters2'] == ol['iters2'][:Rf]).sum()), Rf))
