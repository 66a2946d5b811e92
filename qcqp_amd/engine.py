"""Engine: one QCQPForm resident on one MI355X, driven through the C ABI (include/qcqp_mi.h)."""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _ffi
from .form import RELOP_CODE


class EngineError(Exception):
    """An entry point of libqcqp_mi.so returned an error; `code` is the QCQPMI_E* value (include/qcqp_mi.h)."""
    code = 0

    def __init__(self, message, code=0):
        Exception.__init__(self, message)
        self.code = int(code)


E_UNSUPPORTED = -4     # QCQPMI_EUNSUPPORTED


def _dp(a):
    return a.ctypes.data_as(_ffi.c_dp) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(_ffi.c_ip) if a is not None else None


def _bp(a):
    return a.ctypes.data_as(_ffi.c_bp) if a is not None else None


def device_count():
    return _ffi.lib().qcqpmi_device_count()


class Engine(object):
    KERNEL_EVAL, KERNEL_CD1, KERNEL_CD2, KERNEL_SDR, KERNEL_ADMM = 0, 1, 2, 3, 4

    def __init__(self, form, device=0):
        self.L = _ffi.lib()
        self.n, self.m = int(form.n), int(form.m)
        h = C.c_void_p()
        rc = self.L.qcqpmi_ctx_create(C.byref(h), self.n, self.m, int(device))
        if rc:
            raise EngineError(self.L.qcqpmi_last_error(None).decode())
        self.h = h
        if hasattr(form, 'specs'):     # GeneratedForm: functions synthesised on the device
            for k, g in enumerate(form.specs):
                self._chk(self.L.qcqpmi_set_quad_generated(self.h, k, int(g['seed']), float(g['scale']), float(g['qscale']),
                                                           float(g['diag_add']), float(g['r']), RELOP_CODE[g['relop']]))
        else:
            for k, f in enumerate([form.f0] + list(form.fs)):
                self._set_quad(k, f)
        self._chk(self.L.qcqpmi_finalize(self.h))

    # ------------------------------------------------------------------ plumbing
    def _chk(self, rc):
        if rc:
            raise EngineError(self.L.qcqpmi_last_error(self.h).decode(), rc)

    def close(self):
        if getattr(self, 'h', None):
            self.L.qcqpmi_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _set_quad(self, k, f):
        q = np.ascontiguousarray(f.qarray, dtype=np.float64)
        # (P + P^T)/2 like get_qcqp_form (utilities.py:333, 345): exact no-op for symmetric input; the C ABI
        # rejects a non-symmetric matrix (the one-variable coefficients assume symmetry, utilities.py:99-105)
        if sp.issparse(f.P):
            P = sp.csr_matrix((f.P + f.P.T) / 2.)
            P.sum_duplicates()
            P.sort_indices()
            ptr = np.ascontiguousarray(P.indptr, dtype=np.int64)
            idx = np.ascontiguousarray(P.indices, dtype=np.int64)
            val = np.ascontiguousarray(P.data, dtype=np.float64)
            rc = self.L.qcqpmi_set_quad(self.h, k, 1, _dp(val), _ip(idx), _ip(ptr), len(val), _dp(q),
                                        float(f.r), RELOP_CODE[f.relop])
        else:
            P = np.asarray(f.P, dtype=np.float64)
            P = np.ascontiguousarray((P + P.T) / 2.)
            rc = self.L.qcqpmi_set_quad(self.h, k, 0, _dp(P), None, None, 0, _dp(q), float(f.r),
                                        RELOP_CODE[f.relop])
        self._chk(rc)

    @property
    def separable(self):
        return bool(self.L.qcqpmi_is_separable(self.h))

    # ---------------------------------------------------------------- population
    @property
    def pop_size(self):
        return int(self.L.qcqpmi_pop_size(self.h))

    def upload(self, X):
        """X: (n, R) array, one candidate per column."""
        X = np.asarray(X, dtype=np.float64)
        if X.ndim == 1:
            X = X.reshape(-1, 1)
        assert X.shape[0] == self.n
        Xc = np.ascontiguousarray(X.T)
        self._chk(self.L.qcqpmi_pop_upload(self.h, _dp(Xc), Xc.shape[0]))

    def download(self, R=None):
        R = self.pop_size if R is None else R
        out = np.empty((R, self.n))
        self._chk(self.L.qcqpmi_pop_download(self.h, _dp(out), R))
        return np.ascontiguousarray(out.T)

    def randn(self, R, seed=0, first_index=0):
        self._chk(self.L.qcqpmi_pop_randn(self.h, int(R), int(seed), int(first_index)))

    def sdr_sample(self, mu, F, S, seed=0, first_index=0, Xi=None):
        """x_s = mu + F xi_s for S samples.  mu = F = None draws again from the pair of the previous call (resident)."""
        if mu is None and F is None:
            mu_p = F_p = None
        else:
            mu = np.ascontiguousarray(mu, dtype=np.float64).ravel()
            F = np.ascontiguousarray(F, dtype=np.float64)
            assert F.shape == (self.n, self.n) and mu.size == self.n
            mu_p, F_p = mu, F
        Xc = None
        if Xi is not None:
            Xi = np.asarray(Xi, dtype=np.float64)
            assert Xi.shape == (self.n, S)
            Xc = np.ascontiguousarray(Xi.T)
        self._chk(self.L.qcqpmi_pop_sdr_sample(self.h, _dp(mu_p), _dp(F_p), int(S), int(seed),
                                               int(first_index), _dp(Xc)))

    def sdr_sample_eval(self, mu, F, S, seed=0, first_index=0, want_X=False):
        """S samples x_s = mu + F xi_s drawn AND evaluated in one call (qcqpmi_sdr_sample_eval; qcqp.py:396-401 for S samples):
        returns (f0, maxviol[, X (n, S)]).  No population of S points is laid out: chunks through two reused buffers; sample
        first_index + s is the same point whatever S (keyed normals).  mu = F = None: the pair of the previous call."""
        if mu is None and F is None:
            mu_p = F_p = None
        else:
            mu_p = np.ascontiguousarray(mu, dtype=np.float64).ravel()
            F_p = np.ascontiguousarray(F, dtype=np.float64)
            assert F_p.shape == (self.n, self.n) and mu_p.size == self.n
        S = int(S)
        f0 = np.empty(S)
        mv = np.empty(S)
        Xc = np.empty((S, self.n)) if want_X else None
        self._chk(self.L.qcqpmi_sdr_sample_eval(self.h, _dp(mu_p), _dp(F_p), S, int(seed), int(first_index), _dp(Xc), _dp(f0), _dp(mv)))
        return (f0, mv, np.ascontiguousarray(Xc.T)) if want_X else (f0, mv)

    # ---------------------------------------------------------------- evaluation
    def eval(self, want_F=False):
        R = self.pop_size
        f0 = np.empty(R)
        mv = np.empty(R)
        F = np.empty((self.m + 1, R)) if want_F else None
        self._chk(self.L.qcqpmi_pop_eval(self.h, _dp(f0), _dp(mv), _dp(F)))
        return (f0, mv, F) if want_F else (f0, mv)

    def eval_batch(self, X, want_F=False):
        self.upload(X)
        return self.eval(want_F)

    # -------------------------------------------------------- coordinate descent
    def cd_run(self, phase1=True, num_iters=1000, viol_tol=1e-2, tol=1e-4, seed=0, first_index=0):
        R = self.pop_size
        out = dict(sweeps1=np.zeros(R, dtype=np.int64), sweeps2=np.zeros(R, dtype=np.int64),
                   visits2=np.zeros(R, dtype=np.int64), accepted2=np.zeros(R, dtype=np.int64),
                   ran_phase2=np.zeros(R, dtype=np.uint8), f0=np.empty(R), maxviol=np.empty(R))
        self._chk(self.L.qcqpmi_cd_run(self.h, int(bool(phase1)), int(num_iters), float(viol_tol),
                                       float(tol), int(seed), int(first_index), _ip(out['sweeps1']),
                                       _ip(out['sweeps2']), _ip(out['visits2']), _ip(out['accepted2']),
                                       _bp(out['ran_phase2']), _dp(out['f0']), _dp(out['maxviol'])))
        st1 = np.zeros(R, dtype=np.int32)
        st2 = np.zeros(R, dtype=np.int32)
        self._chk(self.L.qcqpmi_cd_status(self.h, st1.ctypes.data_as(C.POINTER(C.c_int)),
                                          st2.ctypes.data_as(C.POINTER(C.c_int))))
        out['status1'], out['status2'] = st1, st2     # != 0: the reference would raise on this restart (f0 = inf)
        return out

    def cd_stream_reserve(self, K, R):
        """Allocate what cd_stream_run(K, R) needs (population, output staging, winners) ahead of time."""
        self._chk(self.L.qcqpmi_cd_stream_reserve(self.h, int(K), int(R)))

    def cd_stream_run(self, K, R, generate=True, phase1=True, num_iters=1000, viol_tol=1e-2, tol=1e-4, seed=0, seed_stride=1,
                      first_index=0, first_stride=0, select_tol=1e-4, want_best_x=True):
        """K populations of R restarts -- suggest(RANDOM) + improve(COORD_DESCENT) + best point each -- in ONE persistent launch
        (qcqpmi_cd_stream_run, Boolean family).  Population p: seed + p seed_stride, global restart indices first_index +
        p first_stride + [0, R).  Returns the dictionary of cd_run over all K R restarts (population-major) plus
        best_index / best_f0 / best_maxviol (K,) and best_x (K, n)."""
        K, R = int(K), int(R)
        T = K * R
        out = dict(sweeps1=np.zeros(T, dtype=np.int64), sweeps2=np.zeros(T, dtype=np.int64),
                   visits2=np.zeros(T, dtype=np.int64), accepted2=np.zeros(T, dtype=np.int64),
                   ran_phase2=np.zeros(T, dtype=np.uint8), f0=np.empty(T), maxviol=np.empty(T),
                   best_index=np.zeros(K, dtype=np.int64), best_f0=np.empty(K), best_maxviol=np.empty(K),
                   best_x=np.zeros((K, self.n)) if want_best_x else None)
        self._chk(self.L.qcqpmi_cd_stream_run(self.h, K, R, int(bool(generate)), int(bool(phase1)), int(num_iters), float(viol_tol),
                                              float(tol), int(seed), int(seed_stride), int(first_index), int(first_stride),
                                              float(select_tol), _ip(out['sweeps1']), _ip(out['sweeps2']), _ip(out['visits2']),
                                              _ip(out['accepted2']), _bp(out['ran_phase2']), _dp(out['f0']), _dp(out['maxviol']),
                                              _ip(out['best_index']), _dp(out['best_f0']), _dp(out['best_maxviol']),
                                              _dp(out['best_x']) if want_best_x else None))
        st1 = np.zeros(T, dtype=np.int32)
        st2 = np.zeros(T, dtype=np.int32)
        self._chk(self.L.qcqpmi_cd_status(self.h, st1.ctypes.data_as(C.POINTER(C.c_int)), st2.ctypes.data_as(C.POINTER(C.c_int))))
        out['status1'], out['status2'] = st1, st2
        return out

    # the same run in stages (see qcqpmi_cd_run_stage): cd_begin and cd_phase2 only enqueue work on this context's stream
    def cd_begin(self, phase1=True, num_iters=1000, viol_tol=1e-2, tol=1e-4, seed=0, first_index=0):
        """Phase 1 + evaluation + gate of the resident population, asynchronous."""
        self._cd_args = (int(bool(phase1)), int(num_iters), float(viol_tol), float(tol), int(seed), int(first_index))
        self._chk(self.L.qcqpmi_cd_run_stage(self.h, 1, *self._cd_args, None, None, None, None, None, None, None))

    def cd_phase2(self):
        """Launch of phase 2, asynchronous."""
        self._chk(self.L.qcqpmi_cd_run_stage(self.h, 2, *self._cd_args, None, None, None, None, None, None, None))

    def cd_fetch(self):
        """Results of the staged run (blocks until phase 2 has finished): the dictionary of cd_run."""
        R = self.pop_size
        out = dict(sweeps1=np.zeros(R, dtype=np.int64), sweeps2=np.zeros(R, dtype=np.int64),
                   visits2=np.zeros(R, dtype=np.int64), accepted2=np.zeros(R, dtype=np.int64),
                   ran_phase2=np.zeros(R, dtype=np.uint8), f0=np.empty(R), maxviol=np.empty(R))
        self._chk(self.L.qcqpmi_cd_run_stage(self.h, 3, *self._cd_args, _ip(out['sweeps1']),
                                             _ip(out['sweeps2']), _ip(out['visits2']), _ip(out['accepted2']),
                                             _bp(out['ran_phase2']), _dp(out['f0']), _dp(out['maxviol'])))
        st1 = np.zeros(R, dtype=np.int32)
        st2 = np.zeros(R, dtype=np.int32)
        self._chk(self.L.qcqpmi_cd_status(self.h, st1.ctypes.data_as(C.POINTER(C.c_int)),
                                          st2.ctypes.data_as(C.POINTER(C.c_int))))
        out['status1'], out['status2'] = st1, st2
        return out

    # ----------------------------------------------------------------------- ADMM
    def admm_set_eig(self, lmb, Q):
        """lmb: (m, n), Q: (m, n, n) in NumPy eigh layout (Q[k][:, j] = eigenvector j)."""
        lmb = np.ascontiguousarray(lmb, dtype=np.float64)
        Q = np.ascontiguousarray(Q, dtype=np.float64)
        assert lmb.shape == (self.m, self.n) and Q.shape == (self.m, self.n, self.n)
        self._chk(self.L.qcqpmi_admm_set_eig(self.h, _dp(lmb), _dp(Q)))

    def admm_setup(self, method='rocsolver'):
        """Eigenpairs of every constraint matrix computed on the device, in place of admm_set_eig.
        method='rocsolver': batched dsyevd on the dense matrices (any rank)."""
        assert method == 'rocsolver'
        self._chk(self.L.qcqpmi_admm_setup(self.h))

    def admm_set_basis(self, lam, Bv, qhat):
        """Reduced bases of low-rank constraints: lam (m, rp), Bv (m, rp, n) orthonormal rows (zero padded), qhat (m, rp)."""
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        Bv = np.ascontiguousarray(Bv, dtype=np.float64)
        qhat = np.ascontiguousarray(qhat, dtype=np.float64)
        rp = lam.shape[1]
        assert lam.shape == (self.m, rp) and Bv.shape == (self.m, rp, self.n) and qhat.shape == (self.m, rp)
        self._chk(self.L.qcqpmi_admm_set_basis(self.h, rp, _dp(lam), _dp(Bv), _dp(qhat)))

    def admm_set_bracket(self, slo, ehi):
        """Start the multiplier bisection of onecons_qcqp from the given bracket per constraint (utilities.py:176-186;
        -inf / +inf: the reference's doubling search) instead of the one derived from the installed eigenvalues."""
        slo = np.ascontiguousarray(slo, dtype=np.float64)
        ehi = np.ascontiguousarray(ehi, dtype=np.float64)
        assert slo.shape == (self.m,) and ehi.shape == (self.m,)
        self._chk(self.L.qcqpmi_admm_set_bracket(self.h, _dp(slo), _dp(ehi)))

    @staticmethod
    def reference_bracket(lmb):
        """The bracket the reference derives from a FULL eigenvalue list per constraint (utilities.py:176-180): lmb (m, n)
        -> (slo, ehi); round-off eigenvalues of a null space count like any other."""
        lmb = np.asarray(lmb, dtype=np.float64)
        with np.errstate(divide='ignore'):
            inv = -1.0 / lmb
        slo = np.where(lmb > 0, inv, -np.inf).max(axis=1)
        ehi = np.where(lmb < 0, inv, np.inf).min(axis=1)
        return slo, ehi

    def admm_apply_constraints(self, V, shared=True):
        """out[k] = P_k V_k (n x p blocks) on the device; V: (n, p) shared by all constraints or (m, n, p)."""
        V = np.ascontiguousarray(V, dtype=np.float64)
        p = V.shape[-1]
        assert V.shape == ((self.n, p) if shared else (self.m, self.n, p))
        out = np.empty((self.m, self.n, p))
        self._chk(self.L.qcqpmi_admm_apply_constraints(self.h, p, _dp(V), 1 if shared else 0, _dp(out)))
        return out

    def admm_onecons(self, k):
        """onecons_qcqp(z, f_k) (utilities.py:149-196) for every resident point (k = 1..m); returns (n, R)."""
        out = np.empty((self.pop_size, self.n))
        self._chk(self.L.qcqpmi_admm_onecons(self.h, int(k), _dp(out)))
        return np.ascontiguousarray(out.T)

    def admm_zsolver_device(self, rho, max_iter=0):
        """(2 (P0 + rho m I))^-1 formed on the device (Newton-Schulz on the engine's GEMM) and kept for admm_run(rho, None).
        Returns (residual estimate, iterations)."""
        res = np.zeros(1)
        its = np.zeros(1, dtype=np.int64)
        self._chk(self.L.qcqpmi_admm_zsolver_device(self.h, float(rho), int(max_iter), _dp(res), _ip(its)))
        return float(res[0]), int(its[0])

    def p0_lambda_min(self, max_steps=0, tol=1e-12):
        """lambda_min(P0) by Lanczos with device products (qcqp.py:262, 272).  Returns (value, steps)."""
        v = np.zeros(1)
        st = np.zeros(1, dtype=np.int64)
        self._chk(self.L.qcqpmi_p0_lambda_min(self.h, int(max_steps), float(tol), _dp(v), _ip(st)))
        return float(v[0]), int(st[0])

    def admm_run(self, rho, Minv, phase1=True, num_iters=1000, tol=1e-2, viol_lim=1e4):
        """Minv = (2 (P0 + rho m I))^-1 (n, n); None when P0 is diagonal (formed on the device) or after
        admm_zsolver_device(rho)."""
        R = self.pop_size
        if Minv is not None:
            Minv = np.ascontiguousarray(Minv, dtype=np.float64)
            assert Minv.shape == (self.n, self.n)
        out = dict(iters1=np.zeros(R, dtype=np.int64), iters2=np.zeros(R, dtype=np.int64),
                   f0=np.empty(R), maxviol=np.empty(R))
        self._chk(self.L.qcqpmi_admm_run(self.h, int(bool(phase1)), int(num_iters), float(tol),
                                         float(viol_lim), float(rho), _dp(Minv), _ip(out['iters1']),
                                         _ip(out['iters2']), _dp(out['f0']), _dp(out['maxviol'])))
        return out

    def admm_fused(self, enable=True):
        """Fused persistent ADMM kernel on / off (off = the multi-launch path everywhere: the cross-check); 2 = the fused kernel
        with four-wave workgroups, two per compute unit (an experiment kept as a second cross-check)."""
        self._chk(self.L.qcqpmi_admm_fused(self.h, 2 if enable == 2 else (1 if enable else 0)))

    def admm_unit_bases(self, enable=True):
        """Bases of unit vectors (separable constraints): gather / scatter instead of the two consensus GEMMs of an ADMM iteration
        (default on; off = the GEMM path on the same bases: the cross-check)."""
        self._chk(self.L.qcqpmi_admm_unit_bases(self.h, 1 if enable else 0))

    def last_admm_kernel(self):
        """('admm_fused_kernel' | 'admm_multi_launch', workgroups per tile) of the most recent admm_run."""
        cw = C.c_int(0)
        name = self.L.qcqpmi_last_admm_kernel(self.h, C.byref(cw)) or b''
        return name.decode(), int(cw.value)

    def weighted_matrix(self, w):
        """sum_k w_k P_k (n x n) assembled on the device and downloaded."""
        w = np.ascontiguousarray(w, dtype=np.float64)
        S = np.empty((self.n, self.n))
        self._chk(self.L.qcqpmi_weighted_matrix(self.h, _dp(w), _dp(S)))
        return S

    def linear_terms(self):
        """(Q (m+1, n), r (m+1,), relops) as the context holds them (also for device-generated functions)."""
        Q = np.empty((self.m + 1, self.n)); r = np.empty(self.m + 1); rel = []
        rr = C.c_double(0.0); ro = C.c_int(0)
        for k in range(self.m + 1):
            row = np.empty(self.n)
            self._chk(self.L.qcqpmi_get_linear(self.h, k, _dp(row), C.byref(rr), C.byref(ro)))
            Q[k] = row; r[k] = rr.value; rel.append({0: None, 1: '<=', 2: '=='}[ro.value])
        return Q, r, rel

    def eval_parts(self):
        """(quad, lin): quad[k, r] = x_r' P_k x_r + r_k and lin[k, r] = q_k' x_r for the resident population."""
        R = self.pop_size
        quad = np.empty((self.m + 1, R))
        lin = np.empty((self.m + 1, R))
        self._chk(self.L.qcqpmi_pop_eval_parts(self.h, _dp(quad), _dp(lin)))
        return quad, lin

    def weighted_product(self, w):
        """(sum_k w_k P_k) X for the resident population; w has m + 1 entries (objective first).  Returns (n, R)."""
        w = np.ascontiguousarray(w, dtype=np.float64)
        assert w.size == self.m + 1
        out = np.empty((self.pop_size, self.n))
        self._chk(self.L.qcqpmi_pop_weighted_product(self.h, _dp(w), _dp(out)))
        return np.ascontiguousarray(out.T)

    # ------------------------------------------------------------ SDP relaxation
    def sdr_solve_unitdiag(self, Cm, V0=None, max_sweeps=2000, tol=1e-10, seed=0):
        """min <C, X> s.t. diag(X) = 1, X PSD (mixing method on the device).  Returns V (N x 64, unit
        rows, X = V V^T), the objective history and the number of sweeps."""
        Cm = np.ascontiguousarray(Cm, dtype=np.float64)
        N = Cm.shape[0]
        assert Cm.shape == (N, N)
        if V0 is None:
            V0 = np.random.RandomState(seed).randn(N, 64)
            V0 /= np.linalg.norm(V0, axis=1)[:, None]
        V = np.ascontiguousarray(V0, dtype=np.float64).copy()
        hist = np.zeros(max_sweeps + 2)
        sw = C.c_int(0)
        self._chk(self.L.qcqpmi_sdr_solve_unitdiag(self.h, _dp(Cm), N, _dp(V), int(max_sweeps), float(tol), _dp(hist), C.byref(sw)))
        return V, hist[:sw.value + 2], sw.value

    # ------------------------------------------------------------------ selection
    def select_best(self, tol=1e-4, want_x=True):
        idx = np.zeros(1, dtype=np.int64)
        f = np.zeros(1)
        v = np.zeros(1)
        x = np.empty(self.n) if want_x else None
        self._chk(self.L.qcqpmi_select_best(self.h, float(tol), _ip(idx), _dp(f), _dp(v), _dp(x)))
        return int(idx[0]), float(f[0]), float(v[0]), x

    # --------------------------------------------------------------------- timing
    def kernel_ms(self, which):
        ms = np.zeros(1)
        self._chk(self.L.qcqpmi_last_kernel_ms(self.h, int(which), _dp(ms)))
        return float(ms[0])

    def last_cd_kernel(self):
        """Name of the phase-2 kernel the most recent cd_run dispatched to (see qcqpmi_last_cd_kernel)."""
        return (self.L.qcqpmi_last_cd_kernel(self.h) or b'').decode()

    def cd_queue(self, mode=1):
        """Phase-2 scheduling of the Boolean family: 0 = tile-bound (cd_phase2_q_kernel), 1 = restart-level slots refilled from
        a device-side queue (cd_phase2_qs_kernel)."""
        self._chk(self.L.qcqpmi_cd_queue(self.h, int(mode)))

    def cd_life_version(self, version=0):
        """Lifecycle kernel of cd_stream_run: 0 = the faster one for the shape (default), 2 = cd_life_kernel (round 5) wherever it
        applies, 1 = cd_phase2_qs_kernel<lifecycle> (round 4) only."""
        self._chk(self.L.qcqpmi_cd_life_version(self.h, int(version)))

    def cd_set_objective_factor(self, L=None):
        """P0 = L L^T (L: n x r): the lifecycle kernel of cd_stream_run then carries Y = L^T X instead of multiplying with P0
        (qcqpmi_cd_set_objective_factor; qcqp_amd.lowrank.objective_factor finds L).  None removes the factor."""
        if L is None:
            self._chk(self.L.qcqpmi_cd_set_objective_factor(self.h, None, 0))
            return
        L = np.ascontiguousarray(L, dtype=np.float64)
        if L.ndim != 2 or L.shape[0] != self.n:
            raise ValueError('objective factor: expected an n x r matrix')
        self._chk(self.L.qcqpmi_cd_set_objective_factor(self.h, _dp(L), int(L.shape[1])))

    def cd_reference_order(self, enable=True):
        """Coupled constraints: coordinate descent in the reference's summation order (slow, value-for-value comparable with
        the reference at any n; see qcqpmi_cd_reference_order).  No effect on separable problems."""
        self._chk(self.L.qcqpmi_cd_reference_order(self.h, 1 if enable else 0))

    def cd_dense_block_step(self, phase, sweep, block, slack=None, viol_tol=1e-2, tol=1e-4, seed=0, first_index=0, coords=(0, 16)):
        """The unit step of the default dense-constraint path on the resident points: the 16 coordinate visits of `block` in
        sweep `sweep` of phase 1 / phase 2 (coords: the sub-range of the block to visit), from fresh function values
        (qcqpmi_cd_dense_block_step).  slack: the phase-2
        `viol` of qcqp.py:157 per restart (None: the max violation of the resident points)."""
        sl = None
        if slack is not None:
            sl = np.ascontiguousarray(slack, dtype=np.float64)
            assert sl.size == self.pop_size
        self._chk(self.L.qcqpmi_cd_dense_block_step(self.h, int(phase), int(sweep), int(block), int(coords[0]), int(coords[1]), float(viol_tol), float(tol),
                                                    int(seed), int(first_index), _dp(sl) if sl is not None else None))

    def onevar_coeffs(self, coord):
        """QuadraticFunction.get_onevar_func (utilities.py:99-105) on the device: (t2, t1, t0) of every function in the
        coordinate coord[r] of restart r of the resident population.  Returns (R, m + 1, 3)."""
        coord = np.ascontiguousarray(coord, dtype=np.int64)
        assert coord.size == self.pop_size
        out = np.empty((self.pop_size, self.m + 1, 3))
        self._chk(self.L.qcqpmi_onevar_coeffs(self.h, coord.ctypes.data_as(C.POINTER(C.c_int64)), _dp(out)))
        return out

    def dense_chain_mode(self, mode=0):
        """Chain kernel of the dense-constraint path: 0 four waves per restart (default), 1 one wave per restart (the
        round-2 kernel, kept as the cross-check; same points bit for bit)."""
        self._chk(self.L.qcqpmi_dense_chain_mode(self.h, int(mode)))

    def sync(self):
        self._chk(self.L.qcqpmi_sync(self.h))

    # ----------------------------------------------------------------------- comm
    @staticmethod
    def comm_unique_id():
        L = _ffi.lib()
        buf = np.zeros(128, dtype=np.uint8)
        if L.qcqpmi_comm_unique_id(_bp(buf)):
            raise EngineError(L.qcqpmi_last_error(None).decode())
        return buf

    def comm_init(self, rank, world, uid):
        uid = np.ascontiguousarray(uid, dtype=np.uint8)
        self._chk(self.L.qcqpmi_comm_init(self.h, int(rank), int(world), _bp(uid)))
        self._world = int(world)

    def comm_barrier(self):
        self._chk(self.L.qcqpmi_comm_barrier(self.h))

    def comm_allreduce(self, values, op='max'):
        v = np.ascontiguousarray(np.atleast_1d(np.asarray(values, dtype=np.float64)))
        self._chk(self.L.qcqpmi_comm_allreduce(self.h, _dp(v), v.size, 0 if op == 'max' else 1))
        return v

    def comm_allgather(self, arr):
        """All-gather of a contiguous array: returns an array of shape (world,) + arr.shape with every rank's contribution in rank
        order (one ncclAllGather of the raw bytes: integer fields travel exactly)."""
        a = np.ascontiguousarray(arr)
        world = int(self._world) if getattr(self, '_world', None) else 1
        out = np.empty((world,) + a.shape, dtype=a.dtype)
        self._chk(self.L.qcqpmi_comm_allgather(self.h, a.ctypes.data_as(C.c_void_p), a.nbytes, out.ctypes.data_as(C.c_void_p)))
        return out

    def comm_select_best(self, tol=1e-4, index_offset=0):
        idx = np.zeros(1, dtype=np.int64)
        f = np.zeros(1)
        v = np.zeros(1)
        x = np.empty(self.n)
        self._chk(self.L.qcqpmi_comm_select_best(self.h, float(tol), int(index_offset), _ip(idx), _dp(f),
                                                 _dp(v), _dp(x)))
        return int(idx[0]), float(f[0]), float(v[0]), x
