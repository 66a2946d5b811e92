"""Reduced eigenbases of LOW-RANK constraint matrices for the ADMM improve path, without an O(n^3) eigendecomposition
per constraint (the reference caches ``f.eigh = LA.eigh(P)`` of the dense n x n matrix, utilities.py:160-162).

Beamforming-type constraints are  +-(a a' + b b')  (rank 2): onecons_qcqp only ever moves a point inside the span of
the eigenvectors of the nonzero eigenvalues (plus the direction of q outside that span), see csrc/admm.h.  Those few
eigenpairs are found with a randomised range finder whose two passes over the matrices run on the device
(``qcqpmi_admm_apply_constraints``:  Y = P Omega,  Z = P Q); what is left for the host are operations on n x p panels
(p <= 16) in plain NumPy -- Gram-Schmidt, a p x p Jacobi eigensolver -- no LAPACK anywhere.

Every result is verified before it is used: extra probe columns must lie in the range that was found (else the matrix is
not low rank and the caller falls back to full eigendecompositions), and Z - Q (Q'Z) must vanish (span(Q) invariant).
"""
import numpy as np


def _mgs(Y, tol):
    """Batched modified Gram-Schmidt with re-orthogonalisation on (m, n, p) panels; columns that are linearly
    dependent (norm after projection below tol * their original norm) become zero.  Returns Q (m, n, p)."""
    m, n, p = Y.shape
    Q = np.zeros_like(Y)
    for c in range(p):
        v = Y[:, :, c].copy()
        n0 = np.linalg.norm(v, axis=1)
        for _ in range(2):
            for j in range(c):
                v -= np.einsum('kn,kn->k', Q[:, :, j], v)[:, None] * Q[:, :, j]
        nv = np.linalg.norm(v, axis=1)
        keep = nv > tol * np.maximum(n0, 1e-300)
        Q[:, :, c] = np.where(keep[:, None], v / np.where(keep, nv, 1.0)[:, None], 0.0)
    return Q


def jacobi_eigh(T, sweeps=40):
    """Batched cyclic Jacobi eigensolver for symmetric (m, p, p) matrices (NumPy only).  Returns (w (m, p), S (m, p, p))
    with T = S diag(w) S'."""
    A = np.array(T, dtype=np.float64, copy=True)
    m, p, _ = A.shape
    S = np.tile(np.eye(p), (m, 1, 1))
    scale = np.sqrt(np.einsum('kij,kij->k', A, A)) + 1e-300
    offmask = 1.0 - np.eye(p)
    for _ in range(sweeps):
        off = np.sqrt(np.einsum('kij,kij->k', A * offmask, A * offmask))     # summed directly: no cancellation
        if np.all(off <= 1e-15 * scale):
            break
        for a in range(p - 1):
            for b in range(a + 1, p):
                apq = A[:, a, b]
                act = np.abs(apq) > 1e-300
                if not act.any():
                    continue
                tau = (A[:, b, b] - A[:, a, a]) / np.where(act, 2.0 * apq, 1.0)
                t = np.where(tau >= 0, 1.0, -1.0) / (np.abs(tau) + np.sqrt(1.0 + tau * tau))
                t = np.where(act, t, 0.0)
                c = 1.0 / np.sqrt(1.0 + t * t)
                s = t * c
                # A <- J' A J with J the rotation in the (a, b) plane
                Aa, Ab = A[:, :, a].copy(), A[:, :, b].copy()
                A[:, :, a] = c[:, None] * Aa - s[:, None] * Ab
                A[:, :, b] = s[:, None] * Aa + c[:, None] * Ab
                Ra, Rb = A[:, a, :].copy(), A[:, b, :].copy()
                A[:, a, :] = c[:, None] * Ra - s[:, None] * Rb
                A[:, b, :] = s[:, None] * Ra + c[:, None] * Rb
                Sa, Sb = S[:, :, a].copy(), S[:, :, b].copy()
                S[:, :, a] = c[:, None] * Sa - s[:, None] * Sb
                S[:, :, b] = s[:, None] * Sa + c[:, None] * Sb
    return np.einsum('kii->ki', A).copy(), S


def reduced_bases(engine, form, max_rank=8, probes=3, seed=0, tol=1e-10):
    """(lam (m, rp), Bv (m, rp, n), qhat (m, rp), info) for qcqpmi_admm_set_basis, or None when some constraint matrix is
    not of rank <= max_rank (to the accuracy checked).  rp is 2, 4 or 8 when it fits, else the smallest sufficient size."""
    n, m = form.n, form.m
    pb = max_rank + 2                      # basis columns (two more than the rank that is accepted)
    p = pb + probes
    if p > 64 or n < 4 * p:
        return None
    rs = np.random.RandomState(seed)
    Om = rs.randn(n, p)
    Y = engine.admm_apply_constraints(Om, shared=True)            # (m, n, p): the range probe on the device
    Q = _mgs(Y[:, :, :pb], 1e-9)
    # the extra probe columns must already lie in span(Q): otherwise the rank exceeds what was captured
    Yt = Y[:, :, pb:]
    res = Yt - np.einsum('knc,kct->knt', Q, np.einsum('knc,knt->kct', Q, Yt))
    if np.any(np.linalg.norm(res, axis=1) > 1e-8 * (np.linalg.norm(Yt, axis=1) + 1e-300)):
        return None
    if np.any(np.linalg.norm(Q[:, :, max_rank:], axis=1) > 0):   # more than max_rank independent directions
        return None
    Z = engine.admm_apply_constraints(Q, shared=False)            # (m, n, pb): P_k Q_k on the device
    T = np.einsum('knc,knd->kcd', Q, Z)
    T = 0.5 * (T + np.transpose(T, (0, 2, 1)))
    inv_res = Z - np.einsum('knc,kcd->knd', Q, T)                 # span(Q) invariant under P_k ?
    if np.any(np.linalg.norm(inv_res, axis=(1, 2)) > 1e-8 * (np.linalg.norm(Z, axis=(1, 2)) + 1e-300)):
        return None
    w, S = jacobi_eigh(T)
    V = np.einsum('knc,kcd->knd', Q, S)                           # eigenvectors (m, n, pb)
    wmax = np.max(np.abs(w), axis=1)
    keep = np.abs(w) > tol * np.maximum(wmax, 1e-300)[:, None]
    qs = np.array([np.asarray(f.qarray, dtype=np.float64).ravel() for f in form.fs])      # (m, n)
    rank = keep.sum(axis=1)
    rows = []
    for k in range(m):
        idx = np.nonzero(keep[k])[0]
        idx = idx[np.argsort(w[k, idx])]                           # ascending like eigh
        lam_k = list(w[k, idx])
        vec_k = [V[k, :, j] / np.linalg.norm(V[k, :, j]) for j in idx]
        qh_k = [float(v.dot(qs[k])) for v in vec_k]
        qn = qs[k] - sum(h * v for h, v in zip(qh_k, vec_k)) if vec_k else qs[k].copy()
        nq = np.linalg.norm(qn)
        if nq > 1e-13 * (1.0 + np.linalg.norm(qs[k])):
            lam_k.append(0.0); vec_k.append(qn / nq); qh_k.append(float(nq))
        rows.append((lam_k, vec_k, qh_k))
    need = max(len(r[0]) for r in rows)
    rp = 2 if need <= 2 else 4 if need <= 4 else 8 if need <= 8 else need
    lam = np.zeros((m, rp)); Bv = np.zeros((m, rp, n)); qhat = np.zeros((m, rp))
    for k, (lam_k, vec_k, qh_k) in enumerate(rows):
        for j in range(len(lam_k)):
            lam[k, j] = lam_k[j]; Bv[k, j] = vec_k[j]; qhat[k, j] = qh_k[j]
    return lam, Bv, qhat, dict(rank=rank, rp=rp, probes=probes)


def objective_factor(P0, max_rank=288, tol=1e-12):
    """L (n x r, r <= max_rank) with P0 = L L^T to `tol` of the largest |entry| of P0, or None when P0 has no such factor
    (indefinite, or rank above max_rank) -- for qcqpmi_cd_set_objective_factor: the lifecycle kernel then carries L^T X instead
    of multiplying with P0 (a least-squares objective |A x - b|^2 has P0 = A^T A of rank rows(A)).

    Pivoted Cholesky (outer-product form, NumPy only, O(n r^2)): at step k the largest remaining diagonal entry picks the
    column; the factor is VERIFIED against P0 entry by entry before it is returned."""
    P = np.asarray(P0, dtype=np.float64)
    if P.ndim != 2 or P.shape[0] != P.shape[1]:
        return None
    n = P.shape[0]
    scale = float(np.max(np.abs(P))) if n else 0.0
    if not np.isfinite(scale) or scale == 0.0:
        return None
    d = np.diag(P).astype(np.float64).copy()
    if d.min() < -tol * scale:
        return None
    rmax = int(min(max_rank, n))
    L = np.zeros((n, rmax))
    r = 0
    stop = 64.0 * np.finfo(np.float64).eps * scale * max(1.0, np.sqrt(n))
    from ._threads import blas_limit
    with blas_limit():        # (a BLAS pool as wide as the visible CPUs gets a quota-limited container throttled: _threads.py)
        while r < rmax:
            i = int(np.argmax(d))
            if d[i] <= stop:
                break
            col = P[:, i] - L[:, :r] @ L[i, :r]
            L[:, r] = col / np.sqrt(d[i])
            d -= L[:, r] ** 2
            r += 1
        if r == 0 or float(np.max(d)) > max(stop, tol * scale):
            return None                                   # rank above max_rank (or P0 = 0)
        L = np.ascontiguousarray(L[:, :r])
        if float(np.max(np.abs(L @ L.T - P))) > tol * scale:
            return None                                   # not PSD / not symmetric to that accuracy
    return L
