"""SDP relaxation of a QCQP for ``suggest(SDR)`` -- the part the reference hands to cvxpy and an external
SDP solver (``solve_sdr``, qcqp.py:72-97).  Built here for the UNIT-DIAGONAL family: every constraint is
``p x_i^2 + r == 0`` (x_i^2 = d_i > 0) and every coordinate carries exactly one -- Boolean least squares,
MAXCUT, two-way partitioning (BASELINE configs 1-3).  The lifted problem

    minimise <M0, X>   s.t.  X_ii = d_i (i < n),  X_nn = 1,  X PSD,      M0 = [[P0, q0/2], [q0'/2, r0]]

(homogeneous form, utilities.py:66-67) is scaled to unit diagonal and solved on the device by the mixing
method (``qcqpmi_sdr_solve_unitdiag``, csrc/sdr_solve.h).  Other families still need ``X`` from the caller.

The reference's solver is third party => no parity target exists; results are validated by optimality
conditions: ``dual_certificate`` returns the multipliers y and lambda_min(C + diag(y)) (>= 0 at the optimum,
and  -sum(y) + N min(0, lambda_min)  is a rigorous lower bound of the SDP value for any y).
"""
import numpy as np

from .form import QuadraticFunction  # noqa: F401  (type of form.f0 / form.fs)


def _dense(P):
    return np.asarray(P.todense()) if hasattr(P, 'todense') else np.asarray(P, dtype=np.float64)


def unit_diagonal_family(form):
    """d (n,) with x_i^2 = d_i if the form belongs to the family, else None."""
    n = form.n
    d = np.full(n, np.nan)
    for f in form.fs:
        if f.relop != '==':
            return None
        if np.any(np.asarray(f.qarray) != 0.0):
            return None
        if hasattr(f.P, 'tocoo'):            # scipy sparse: never densify (n matrices of n x n)
            Pc = f.P.tocoo()
            keep = Pc.data != 0.0
            rows, cols, vals = Pc.row[keep], Pc.col[keep], Pc.data[keep]
        else:
            P = np.asarray(f.P)
            rows, cols = np.nonzero(P)
            vals = P[rows, cols]
        if len(vals) != 1 or rows[0] != cols[0]:
            return None
        i = int(rows[0])
        p = float(vals[0])
        di = -f.r / p
        if not (di > 0.0) or not np.isnan(d[i]):
            return None
        d[i] = di
    if np.any(np.isnan(d)):
        return None
    return d


def lifted_cost(form, d):
    """C (N x N, N = n + 1) of the unit-diagonal problem in y = x / sqrt(d), and the scaling."""
    n = form.n
    s = np.sqrt(d)
    P0 = _dense(form.f0.P)
    P0 = 0.5 * (P0 + P0.T)
    q0 = np.asarray(form.f0.qarray, dtype=np.float64).ravel()
    C = np.zeros((n + 1, n + 1))
    C[:n, :n] = P0 * np.outer(s, s)
    C[:n, n] = C[n, :n] = 0.5 * q0 * s
    C[n, n] = form.f0.r
    return C, np.append(s, 1.0)


def dual_certificate(C, V):
    """y_i = -v_i . (C v)_i  (stationarity multipliers of diag(X) = 1), S = C + diag(y).
    Returns (y, lambda_min(S), rigorous lower bound -sum(y) + N min(0, lambda_min))."""
    G = C.dot(V)
    y = -np.einsum('ik,ik->i', V, G)
    S = C + np.diag(y)
    lmin = float(np.linalg.eigvalsh(0.5 * (S + S.T))[0])
    return y, lmin, float(-y.sum() + C.shape[0] * min(0.0, lmin))


def solve_sdr(engine, form, max_sweeps=5000, tol=1e-11, seed=0):
    """Returns (X, bound, info) like the reference's solve_sdr returns (X, bound): X is the lifted
    (n+1) x (n+1) solution in the ORIGINAL variables, bound = <M0, X> (minimise form)."""
    d = unit_diagonal_family(form)
    if d is None:
        return None
    C, sc = lifted_cost(form, d)
    V, hist, sweeps = engine.sdr_solve_unitdiag(C, max_sweeps=max_sweeps, tol=tol, seed=seed)
    Y = V.dot(V.T)
    X = Y * np.outer(sc, sc)
    # rotate so that the homogenising coordinate is +1 exactly (X_nn = 1 already; sign of the last column
    # is fixed by the optimisation itself)
    bound = float(hist[-1])
    return X, bound, dict(V=V, C=C, hist=hist, sweeps=sweeps, scale=sc)
