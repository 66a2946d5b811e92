"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the Suggest-and-Improve hot path.

ctypes front-end of ``oracle/libqcqp_oracle.so`` (plain-C restatement of
cvxgrp/qcqp's hot path, see qcqp_oracle.h) plus the few NumPy-level pieces
that the reference itself does in NumPy/LAPACK (eigendecompositions, the
z-update factorisation, the SDR sampling tail).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.  The product (``qcqp_amd``)
never does: it must fail loudly when the HIP library is missing.

Pinned against golden vectors captured from the reference
(``tests/golden/*.npz``, generator ``tools/gen_golden.py``) by
``tests/test_oracle_golden.py``.

Reference citations are relative to /root/reference/.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RELOP = {None: 0, '<=': 1, '==': 2}
RNG_MT, RNG_KEYED = 0, 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


def build(force=False):
    so = os.path.join(_HERE, 'libqcqp_oracle.so')
    src = os.path.join(_HERE, 'qcqp_oracle.c')
    hdr = os.path.join(_HERE, 'qcqp_oracle.h')
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(
            os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libqcqp_oracle.so'],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_prob_new.restype = C.c_void_p
        L.orc_prob_new.argtypes = [C.c_int64, C.c_int64]
        L.orc_prob_free.argtypes = [C.c_void_p]
        L.orc_prob_set.argtypes = [C.c_void_p, C.c_int64, C.c_int64, _ip, _ip, _dp, _dp,
                                   C.c_double, C.c_int]
        L.orc_rng_new.restype = C.c_void_p
        L.orc_rng_new.argtypes = [C.c_int, C.c_uint64]
        L.orc_rng_free.argtypes = [C.c_void_p]
        L.orc_rng_mt_set.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int]
        L.orc_rng_mt_get.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.orc_rng_set_restart.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_rng_set_ctx.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_rng_uniform.restype = C.c_double
        L.orc_rng_uniform.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.orc_rng_choice.restype = C.c_int64
        L.orc_rng_choice.argtypes = [C.c_void_p, C.c_int64]
        L.orc_rng_draws.restype = C.c_uint64
        L.orc_rng_draws.argtypes = [C.c_void_p]
        L.orc_philox4x32.argtypes = [C.POINTER(C.c_uint32)] * 3
        L.orc_keyed_normal.restype = C.c_double
        L.orc_keyed_normal.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_generated_eval.restype = None
        L.orc_generated_eval.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                         _dp, C.c_int64, _dp]
        L.orc_eval.restype = C.c_double
        L.orc_eval.argtypes = [C.c_void_p, C.c_int64, _dp]
        L.orc_violation.restype = C.c_double
        L.orc_violation.argtypes = [C.c_void_p, C.c_int64, _dp]
        L.orc_max_violation.restype = C.c_double
        L.orc_max_violation.argtypes = [C.c_void_p, _dp]
        L.orc_better.argtypes = [C.c_void_p, _dp, _dp, C.c_double]
        L.orc_eval_batch.argtypes = [C.c_void_p, _dp, C.c_int64, _dp, _dp, _dp]
        L.orc_onevar_coeffs.argtypes = [C.c_void_p, C.c_int64, _dp, C.c_int64, _dp]
        L.orc_feasible_intervals.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int,
                                             C.c_double, C.c_double, _dp]
        L.orc_onevar_qcqp.argtypes = [C.c_double, C.c_double, C.c_double, _dp,
                                      C.POINTER(C.c_int), C.c_int64, C.c_double, C.c_void_p,
                                      _dp, _dp, C.c_int64, _ip]
        for nm in ('orc_cd_phase1', 'orc_cd_phase2'):
            getattr(L, nm).argtypes = [C.c_void_p, _dp, C.c_int64, C.c_double, C.c_double,
                                       C.c_void_p, _ip]
        L.orc_improve_cd.argtypes = [C.c_void_p, _dp, C.c_int64, C.c_double, C.c_double,
                                     C.c_int, C.c_void_p, _ip, _ip]
        L.orc_cd_trace.restype = None
        L.orc_cd_trace.argtypes = [_dp, C.c_int64]
        L.orc_cd_trace_len.restype = C.c_int64
        L.orc_cd_visit_limit.restype = None
        L.orc_cd_visit_limit.argtypes = [C.c_int64]
        L.orc_cd_visits.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_void_p]
        L.orc_cd_trace_len.argtypes = []
        L.orc_cd_phase2_incremental.argtypes = [C.c_void_p, _dp, _dp, C.c_int64, C.c_double, C.c_void_p, _ip]
        L.orc_onecons.argtypes = [C.c_void_p, C.c_int64, _dp, _dp, _dp, C.c_double, _dp]
        L.orc_admm_phase1.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_double, C.c_int64, _ip]
        L.orc_admm_phase2.argtypes = [C.c_void_p, _dp, C.c_double, _dp, _dp, _dp, C.c_double,
                                      C.c_int64, C.c_double, _ip]
        _LIB = L
    return _LIB


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _vec(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).ravel())


class Rng:
    """MT mode mirrors the global ``np.random`` state call-for-call; KEYED mode is the
    counter-based Philox stream shared with the HIP engine."""

    def __init__(self, mode=RNG_MT, seed=0, from_numpy_global=False):
        self.h = lib().orc_rng_new(mode, seed)
        self.mode = mode
        if from_numpy_global:
            self.pull_numpy()

    def pull_numpy(self):
        st = np.random.get_state()
        key = np.ascontiguousarray(st[1], dtype=np.uint32)
        lib().orc_rng_mt_set(self.h, key.ctypes.data_as(C.POINTER(C.c_uint32)), int(st[2]))

    def push_numpy(self):
        key = np.zeros(624, dtype=np.uint32)
        pos = C.c_int(0)
        lib().orc_rng_mt_get(self.h, key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos))
        st = np.random.get_state()
        np.random.set_state((st[0], key, pos.value, st[3], st[4]))

    def set_ctx(self, coord, sweep_tag, it):
        """Keyed mode: the counter words (coordinate, sweep tag, bisection iteration) of the next draw."""
        lib().orc_rng_set_ctx(self.h, int(coord), int(sweep_tag), int(it))

    def set_restart(self, r):
        lib().orc_rng_set_restart(self.h, int(r))

    def uniform(self, lo, hi):
        return lib().orc_rng_uniform(self.h, lo, hi)

    def choice(self, k):
        return lib().orc_rng_choice(self.h, k)

    @property
    def draws(self):
        return lib().orc_rng_draws(self.h)

    def __del__(self):
        try:
            lib().orc_rng_free(self.h)
        except Exception:
            pass


class Problem:
    """QCQPForm restatement (utilities.py:122-130).  ``funcs`` is a list of
    ``(P, q, r, relop)`` with f0 first (relop None); P dense ndarray or scipy sparse."""

    def __init__(self, funcs):
        f0 = funcs[0]
        self.n = int(np.asarray(f0[1]).size)
        self.m = len(funcs) - 1
        self.funcs = []
        self.h = lib().orc_prob_new(self.n, self.m)
        for k, (P, q, r, relop) in enumerate(funcs):
            Pc = sp.csr_matrix(P, dtype=np.float64)
            Pc.sum_duplicates()
            Pc.sort_indices()
            ptr = np.ascontiguousarray(Pc.indptr, dtype=np.int64)
            idx = np.ascontiguousarray(Pc.indices, dtype=np.int64)
            val = np.ascontiguousarray(Pc.data, dtype=np.float64)
            qv = _vec(q)
            assert qv.size == self.n and Pc.shape == (self.n, self.n)
            rc = lib().orc_prob_set(self.h, k, len(val), _i(ptr), _i(idx), _d(val), _d(qv),
                                    float(r), RELOP[relop])
            assert rc == 0
            self.funcs.append((Pc, qv, float(r), relop))
        self._eig = None

    def __del__(self):
        try:
            lib().orc_prob_free(self.h)
        except Exception:
            pass

    # ---- utilities.py:49-62, 133-146
    def eval(self, k, x):
        return lib().orc_eval(self.h, k, _d(_vec(x)))

    def violation(self, k, x):
        return lib().orc_violation(self.h, k, _d(_vec(x)))

    def max_violation(self, x):
        return lib().orc_max_violation(self.h, _d(_vec(x)))

    def better(self, x1, x2, tol=1e-4):
        """Returns 1 / 2 = which argument QCQPForm.better returns."""
        return lib().orc_better(self.h, _d(_vec(x1)), _d(_vec(x2)), tol)

    def eval_batch(self, X, want_F=False):
        """X: (n, S) array, one candidate per column."""
        X = np.asarray(X, dtype=np.float64)
        S = X.shape[1]
        Xc = np.ascontiguousarray(X.T)  # sample-major
        f0 = np.zeros(S)
        mv = np.zeros(S)
        F = np.zeros((self.m + 1, S)) if want_F else None
        lib().orc_eval_batch(self.h, _d(Xc), S, _d(f0), _d(mv), _d(F) if want_F else None)
        return (f0, mv, F) if want_F else (f0, mv)

    # ---- utilities.py:99-105
    def onevar_coeffs(self, k, x, c):
        out = np.zeros(3)
        lib().orc_onevar_coeffs(self.h, k, _d(_vec(x)), c, _d(out))
        return out

    # ---- qcqp.py:100-192
    def cd_phase1(self, x, num_iters=1000, viol_tol=1e-2, tol=1e-4, rng=None):
        return self._cd('orc_cd_phase1', x, num_iters, viol_tol, tol, rng)

    def cd_phase2(self, x, num_iters=1000, viol_tol=1e-2, tol=1e-4, rng=None):
        return self._cd('orc_cd_phase2', x, num_iters, viol_tol, tol, rng)

    def cd_phase2_incremental(self, x, num_iters=1000, tol=1e-4, rng=None):
        """Optimised CPU baseline of phase 2 (incremental gradient, qcqp.py:152-178 semantics); separable
        constraints only.  Same one-variable solver as cd_phase2, O(n) per accepted move."""
        if getattr(self, '_P0d', None) is None:
            self._P0d = np.ascontiguousarray(self.funcs[0][0].toarray(), dtype=np.float64)
        x = _vec(x).copy()
        stats = np.zeros(3, dtype=np.int64)
        rng = rng or Rng(RNG_MT, 0, from_numpy_global=True)
        rc = lib().orc_cd_phase2_incremental(self.h, _d(self._P0d), _d(x), num_iters, tol, rng.h, _i(stats))
        if rc:
            raise RuntimeError('oracle orc_cd_phase2_incremental failed rc=%d' % rc)
        return x, stats

    def _cd(self, fn, x, num_iters, viol_tol, tol, rng):
        x = _vec(x).copy()
        stats = np.zeros(3, dtype=np.int64)
        rng = rng or Rng(RNG_MT, 0, from_numpy_global=True)
        rc = getattr(lib(), fn)(self.h, _d(x), num_iters, viol_tol, tol, rng.h, _i(stats))
        if rc:
            raise RuntimeError('oracle %s failed rc=%d' % (fn, rc))
        return x, stats

    def improve_cd(self, x, num_iters=1000, viol_tol=1e-2, tol=1e-4, phase1=True, rng=None):
        x = _vec(x).copy()
        s1 = np.zeros(3, dtype=np.int64)
        s2 = np.zeros(3, dtype=np.int64)
        rng = rng or Rng(RNG_MT, 0, from_numpy_global=True)
        rc = lib().orc_improve_cd(self.h, _d(x), num_iters, viol_tol, tol, int(phase1), rng.h,
                                  _i(s1), _i(s2))
        if rc:
            raise RuntimeError('oracle improve_cd failed rc=%d' % rc)
        return x, s1, s2

    def improve_cd_traced(self, x, num_iters=1000, viol_tol=1e-2, tol=1e-4, rng=None):
        """improve_coord_descent (qcqp.py:181-192) with every intermediate state: returns (x, stats1, stats2, tr1, tr2,
        slack2) -- tr1 / tr2 = the value of x[i] after each coordinate visit of phase 1 / phase 2 in visiting order (visit
        v is coordinate v mod n of sweep v // n), slack2 = the `viol` phase 2 fixes at its start (qcqp.py:157), None if
        the gate (qcqp.py:189) kept phase 2 from running.  The state after any visit is x0 with the recorded values
        applied in order: what the teacher-forced parity tests of the dense path feed the engine."""
        L = lib()
        n = self.n
        rng = rng or Rng(RNG_MT, 0, from_numpy_global=True)
        x = _vec(x).copy()
        buf = np.zeros(max(1, num_iters * n))
        s1 = np.zeros(3, dtype=np.int64)
        s2 = np.zeros(3, dtype=np.int64)
        L.orc_cd_trace(_d(buf), buf.size)
        try:
            rc = L.orc_cd_phase1(self.h, _d(x), num_iters, viol_tol, tol, rng.h, _i(s1))
            k1 = int(L.orc_cd_trace_len())
        finally:
            L.orc_cd_trace(None, 0)
        if rc:
            raise RuntimeError('oracle orc_cd_phase1 failed rc=%d' % rc)
        tr1 = buf[:k1].copy()
        tr2, slack2 = np.zeros(0), None
        mv = self.max_violation(x)
        if mv < viol_tol:
            slack2 = mv
            L.orc_cd_trace(_d(buf), buf.size)
            try:
                rc = L.orc_cd_phase2(self.h, _d(x), num_iters, viol_tol, tol, rng.h, _i(s2))
                k2 = int(L.orc_cd_trace_len())
            finally:
                L.orc_cd_trace(None, 0)
            if rc:
                raise RuntimeError('oracle orc_cd_phase2 failed rc=%d' % rc)
            tr2 = buf[:k2].copy()
        return x, s1, s2, tr1, tr2, slack2

    def cd_visits(self, phase, x, sweep, i0, count, slack2=0.0, viol_tol=1e-2, tol=1e-4, rng=None):
        """`count` coordinate visits i0, i0 + 1, ... of sweep `sweep` of phase 1 / phase 2 from the state x (phase 2: with the
        slack `slack2` fixed at its start, qcqp.py:157): the reference's answer for ONE block from an arbitrary state."""
        rng = rng or Rng(RNG_MT, 0, from_numpy_global=True)
        x = _vec(x).copy()
        rc = lib().orc_cd_visits(self.h, _d(x), int(phase), int(sweep), int(i0), int(count), float(slack2), viol_tol, tol, rng.h)
        if rc:
            raise RuntimeError('oracle orc_cd_visits failed rc=%d' % rc)
        return x

    def cd_phase_traced(self, phase, x, visits, viol_tol=1e-2, tol=1e-4, rng=None):
        """The first `visits` coordinate visits of coord_descent_phase1 (qcqp.py:101-149) or coord_descent_phase2 (qcqp.py:152-178)
        from x: returns (x after them, the value of x[i] after each visit, the slack phase 2 fixes at its start -- None for
        phase 1).  For problems whose sweep costs minutes on the host (257 dense 1024 x 1024 functions: 0.13 s per visit)."""
        L = lib()
        rng = rng or Rng(RNG_MT, 0, from_numpy_global=True)
        x = _vec(x).copy()
        buf = np.zeros(max(1, int(visits)))
        st = np.zeros(3, dtype=np.int64)
        slack = None if phase == 1 else self.max_violation(x)
        L.orc_cd_trace(_d(buf), buf.size)
        L.orc_cd_visit_limit(int(visits))
        try:
            rc = (L.orc_cd_phase1 if phase == 1 else L.orc_cd_phase2)(self.h, _d(x), 1000, viol_tol, tol, rng.h, _i(st))
            k = int(L.orc_cd_trace_len())
        finally:
            L.orc_cd_visit_limit(-1)
            L.orc_cd_trace(None, 0)
        if rc:
            raise RuntimeError('oracle cd phase %d failed rc=%d' % (phase, rc))
        return x, buf[:k].copy(), slack

    # ---- utilities.py:149-196
    def eig(self):
        """Per-constraint eigh of sym(P) -- LAPACK through NumPy, like the reference
        (utilities.py:160-162).  Returns (lmb (m,n), Q (m,n,n))."""
        if self._eig is None:
            lm = np.zeros((self.m, self.n))
            Q = np.zeros((self.m, self.n, self.n))
            for k in range(self.m):
                P = self.funcs[k + 1][0]
                Ps = np.asarray(((P + P.T) / 2.).todense())
                lm[k], Q[k] = np.linalg.eigh(Ps)
            self._eig = (np.ascontiguousarray(lm), np.ascontiguousarray(Q))
        return self._eig

    def onecons(self, k, z, lmb=None, Q=None, tol=1e-6):
        if lmb is None:
            L, QQ = self.eig()
            lmb, Q = L[k - 1], QQ[k - 1]
        out = np.zeros(self.n)
        steps = lib().orc_onecons(self.h, k, _d(_vec(z)), _d(_vec(lmb)),
                                  _d(np.ascontiguousarray(Q, dtype=np.float64)), tol, _d(out))
        return out, steps

    # ---- qcqp.py:195-285
    def admm_phase1(self, x0, tol=1e-2, num_iters=1000):
        lm, Q = self.eig()
        z = _vec(x0).copy()
        it = np.zeros(1, dtype=np.int64)
        lib().orc_admm_phase1(self.h, _d(z), _d(lm), _d(Q), tol, num_iters, _i(it))
        return z, int(it[0])

    def admm_phase2(self, x0, rho, tol=1e-2, num_iters=1000, viol_lim=1e4):
        lm, Q = self.eig()
        P0 = np.asarray(self.funcs[0][0].todense())
        chol = np.ascontiguousarray(np.linalg.cholesky(2. * (P0 + rho * self.m * np.eye(self.n))))
        x = _vec(x0).copy()
        it = np.zeros(1, dtype=np.int64)
        lib().orc_admm_phase2(self.h, _d(x), rho, _d(lm), _d(Q), _d(chol), tol, num_iters,
                              viol_lim, _i(it))
        return x, int(it[0])

    def auto_rho(self):
        """qcqp.py:271-277."""
        lmb0 = np.linalg.eigh(np.asarray(self.funcs[0][0].todense()))[0]
        lmb_min = np.min(lmb0)
        rho = 2. * (1. - lmb_min) / self.m if lmb_min < 0 else 1. / self.m
        return rho * 50.

    def improve_admm(self, x0, num_iters=1000, viol_lim=1e4, tol=1e-2, rho=None, phase1=True):
        x0 = _vec(x0)
        if rho is not None:
            lmb_min = np.min(np.linalg.eigh(np.asarray(self.funcs[0][0].todense()))[0])
            if lmb_min + self.m * rho < 0:
                raise Exception("rho parameter is too small, need at least %.3f." % rho)
        else:
            rho = self.auto_rho()
        if phase1:
            z1, _ = self.admm_phase1(x0, tol, num_iters)
            x1 = x0 if self.better(x0, z1) == 1 else z1
        else:
            x1 = x0
        z2, _ = self.admm_phase2(x1, rho, tol, num_iters, viol_lim)
        return x1 if self.better(x1, z2) == 1 else z2


def feasible_intervals(p, q, r, relop, s=0., tol=1e-4):
    out = np.zeros(8)
    c = lib().orc_feasible_intervals(p, q, r, RELOP[relop], s, tol, _d(out))
    return [(out[2 * i], out[2 * i + 1]) for i in range(c)]


def onevar_qcqp(f0, fs, s, rng=None):
    """f0 = (p,q,r); fs = list of (p,q,r,relop).  Returns (x or None, C)."""
    mf = len(fs)
    fs3 = np.zeros(max(3 * mf, 1))
    rel = (C.c_int * max(mf, 1))()
    for k, f in enumerate(fs):
        fs3[3 * k:3 * k + 3] = f[:3]
        rel[k] = RELOP[f[3]]
    rng = rng or Rng(RNG_MT, 0, from_numpy_global=True)
    x = C.c_double(0.)
    cap = 4 * mf + 2
    Cout = np.zeros(2 * cap)
    nC = np.zeros(1, dtype=np.int64)
    got = lib().orc_onevar_qcqp(f0[0], f0[1], f0[2], _d(fs3), rel, mf, s, rng.h, C.byref(x),
                                _d(Cout), cap, _i(nC))
    Cl = [(Cout[2 * i], Cout[2 * i + 1]) for i in range(int(nC[0]))]
    if got < 0:
        raise RuntimeError('oracle onevar_qcqp: reference would raise (rc=%d)' % got)
    return (x.value if got else None), Cl


def philox4x32(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32(c, k, o)
    return list(o)


def keyed_normal(seed, restart, elem):
    return lib().orc_keyed_normal(seed, restart, elem)


def generated_eval(spec, k, n, X):
    """f_k(x) for the columns of X (n, S) of a device-generated function (problems.GeneratedForm.specs[k]) without
    materialising P_k: n (n + 1) / 2 keyed normals, ~1 s per function at n = 4096."""
    X = np.asarray(X, dtype=np.float64)
    Xc = np.ascontiguousarray(X.T)
    out = np.zeros(Xc.shape[0])
    lib().orc_generated_eval(int(spec['seed']), int(k), int(n), float(spec['scale']), float(spec['qscale']),
                             float(spec['diag_add']), float(spec['r']), _d(Xc), Xc.shape[0], _d(out))
    return out


def keyed_normal_matrix(seed, n, R, first_index=0):
    """(n, R) matrix of the keyed normals: column r = restart/sample first_index + r."""
    out = np.empty((n, R))
    L = lib()
    for r in range(R):
        for j in range(n):
            out[j, r] = L.orc_keyed_normal(seed, first_index + r, j)
    return out


# ---- QCQP.suggest SDR tail (qcqp.py:394-401), NumPy level like the reference --------------

def sdr_mu_sigma(X, eps=1e-8, compat=True):
    """mu, Sigma from the lifted solution X ((n+1) x (n+1)).  compat=True reproduces the
    reference's row-broadcast `mu*mu.T` (1-D mu => element-wise square);
    compat=False is the intended X - mu mu^T + eps I."""
    X = np.asarray(X)
    n = X.shape[0] - 1
    mu = np.asarray(X[:-1, -1]).flatten()
    if compat:
        Sigma = X[:-1, :-1] - mu * mu.T + eps * np.eye(n)
    else:
        Sigma = X[:-1, :-1] - np.outer(mu, mu) + eps * np.eye(n)
    return mu, Sigma


def suggest_sdr_sample(prob, mu, Sigma, maximize=False):
    """One np.random.multivariate_normal draw + (f, v) (qcqp.py:396-401); consumes the
    global NumPy RNG exactly like the reference."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        x = np.random.multivariate_normal(mu, Sigma)
    f = prob.eval(0, x)
    if maximize:
        f *= -1
    return x, f, prob.max_violation(x)
