"""Timing + sanity of the other BASELINE configs at full size (cfg3 SDR sampling, cfg4 ADMM)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from qcqp_amd import problems
from qcqp_amd.engine import Engine
from qcqp_amd.form import QCQPForm

which = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
if which == 'cfg3':
    n, S = 2000, 8192
    funcs, maxi, ex = problems.maxcut(n, 0.5, seed=1)
    e = Engine(QCQPForm.from_arrays(funcs))
    # the SDP relaxation, solved by the engine itself (mixing method, csrc/sdr_solve.h)
    from qcqp_amd import sdr
    form = QCQPForm.from_arrays(funcs)
    t0 = time.time(); Xs, bound, info = sdr.solve_sdr(e, form, max_sweeps=400, tol=1e-9); tsd = time.time() - t0
    t0 = time.time(); y, lmin, lower = sdr.dual_certificate(info['C'], info['V']); tce = time.time() - t0
    print('cfg3: SDP relaxation n=%d: %d sweeps in %.2f s; SDP cut bound %.1f; dual certificate lambda_min %.2e (host eigvalsh %.1f s); '
          'rigorous bound %.1f' % (n, info['sweeps'], tsd, -bound, lmin, tce, -lower))
    Sigma = Xs[:n, :n] + 1e-8 * np.eye(n)
    t0 = time.time(); w, U = np.linalg.eigh(Sigma); F = U * np.sqrt(np.maximum(w, 0.0)); tch = time.time() - t0
    mu = np.zeros(n)
    e.sdr_sample(mu, F, S, seed=3)          # warm-up (includes factor upload)
    t0 = time.time(); e.sdr_sample(mu, F, S, seed=4); t1 = time.time()
    f0, mv = e.eval(); t2 = time.time()
    print('cfg3: host factor (eigh) %.3f s; sample call %.1f ms (kernel %.3f ms); eval call %.1f ms (kernel %.3f ms)'
          % (tch, 1e3 * (t1 - t0), e.kernel_ms(3), 1e3 * (t2 - t1), e.kernel_ms(0)))
    X = e.download()
    # properties: sample covariance ~ Sigma, objective identity f0 = x'P0x + r0, cut value of sign rounding
    C = X.dot(X.T) / S
    print('   cov error (rel fro) %.3e' % (np.linalg.norm(C - Sigma) / np.linalg.norm(Sigma)))
    P0, r0 = funcs[0][0], funcs[0][2]
    chk = np.einsum('is,ij,js->s', X[:, :16], P0, X[:, :16]) + r0
    print('   f0 check (16 samples) rel err %.2e' % np.max(np.abs(chk - f0[:16]) / np.abs(chk)))
    xs = np.sign(X); cuts = -(np.einsum('is,ij,js->s', xs[:, :256], P0, xs[:, :256]) + r0)
    print('   GW-rounded cut (256 samples): mean %.0f max %.0f; edges %d; best/SDP bound %.4f (GW guarantee 0.878 in expectation)'
          % (cuts.mean(), cuts.max(), ex['W'].sum() / 2, cuts.max() / -bound))
    flops = 2.0 * n * n * S
    print('   sampling %.1f TFLOP/s, eval %.1f TFLOP/s' % (flops / e.kernel_ms(3) / 1e9, flops / e.kernel_ms(0) / 1e9))
elif which in ('cfg5', 'cdbeam'):
    # coordinate descent with constraints that couple coordinates (dense path): reduced cfg5 / cfg4's family
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    if which == 'cfg5':
        n = int(sys.argv[4]) if len(sys.argv) > 4 else 512
        m = int(sys.argv[5]) if len(sys.argv) > 5 else 128
        funcs, _, _ = problems.dense_indefinite(n, m, seed=7)
        x0s = 0.1
    else:
        nant = int(sys.argv[4]) if len(sys.argv) > 4 else 512
        funcs, _, _ = problems.beamforming(nant, 16, 64, seed=1)
        n, m, x0s = 2 * nant, 80, 1.0
    t0 = time.time(); e = Engine(QCQPForm.from_arrays(funcs)); t1 = time.time()
    rs = np.random.RandomState(0)
    e.upload(x0s * rs.randn(n, R))
    e.eval(); e.sync()
    ta = time.time(); f0, mv = e.eval(); tb = time.time()
    fl = 2.0 * (m + 1) * n * n
    print('%s (n=%d, m=%d, R=%d): engine %.1f s; eval call %.1f ms (kernels %.3f ms = %.1f TFLOP/s); start maxviol median %.3g'
          % (which, n, m, R, t1 - t0, 1e3 * (tb - ta), e.kernel_ms(0), fl * R / e.kernel_ms(0) / 1e9, np.median(mv)))
    ta = time.time(); out = e.cd_run(phase1=True, num_iters=iters, seed=1); tb = time.time()
    s1, s2 = out['sweeps1'].sum(), out['sweeps2'].sum()
    print('   cd_run(num_iters=%d) %.2f s: phase-1 restart-sweeps %d (%.1f ms), phase-2 restart-sweeps %d (%.1f ms); ran phase 2: %d; '
          'feasible %d; f0 median %.4g' % (iters, tb - ta, s1, e.kernel_ms(1), s2, e.kernel_ms(2), out['ran_phase2'].sum(),
                                          (out['maxviol'] < 1e-2).sum(), np.median(out['f0'])))
    for nm, sw, ms in (('phase 1', s1, e.kernel_ms(1)), ('phase 2', s2, e.kernel_ms(2))):
        if sw:
            print('   %s: %.1f restart-sweeps/s, %.2f TFLOP/s algorithmic (2 (m+1) n^2 per restart-sweep)'
                  % (nm, sw / ms * 1e3, fl * sw / ms / 1e9))
else:
    nant, mh, l, R = 512, 16, 64, int(sys.argv[2]) if len(sys.argv) > 2 else 128
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    funcs, _, _ = problems.beamforming(nant, mh, l, seed=1)
    n, m = 2 * nant, mh + l
    t0 = time.time()
    e = Engine(QCQPForm.from_arrays(funcs)); t1 = time.time()
    if len(sys.argv) > 4 and sys.argv[4] == 'host':
        lm = np.zeros((m, n)); Q = np.zeros((m, n, n))
        for k in range(m):
            lm[k], Q[k] = np.linalg.eigh(np.asarray(funcs[k + 1][0]))
        t2 = time.time()
        e.admm_set_eig(lm, Q); t3 = time.time()
    else:           # eigendecompositions on the device (rocSOLVER batched dsyevd)
        t2 = time.time()
        e.admm_setup(); e.sync(); t3 = time.time()
    rho = 1.0
    Minv = np.linalg.inv(2. * (np.eye(n) + rho * m * np.eye(n)))
    e.randn(R, seed=1)
    e.admm_run(rho, Minv, phase1=True, num_iters=2)   # warm-up: rocBLAS initialisation
    e.randn(R, seed=1)
    ta = time.time(); out = e.admm_run(rho, Minv, phase1=True, num_iters=iters); tb = time.time()
    its = out['iters1'].sum() + out['iters2'].sum()
    print('cfg4 (n=%d, m=%d, R=%d): engine %.1f s, host eigh %.1f s, set_eig / device setup %.1f s; admm_run(num_iters=%d) %.2f s; '
          '%d restart-iterations -> %.1f restart-iterations/s; secular kernel %.2f ms'
          % (n, m, R, t1 - t0, t2 - t1, t3 - t2, iters, tb - ta, its, its / (tb - ta), e.kernel_ms(4)))
    print('   f0 range %.3f..%.3f, maxviol max %.3e' % (out['f0'].min(), out['f0'].max(), out['maxviol'].max()))
