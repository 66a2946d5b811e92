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
    if not hasattr(form, 'fs'):          # GeneratedForm: dense by construction
        return None
    d = unit_diagonal_family(form)
    if d is None:
        return None
    C, sc = lifted_cost(form, d)
    V, hist, sweeps = engine.sdr_solve_unitdiag(C, max_sweeps=max_sweeps, tol=tol, seed=seed)
    Y = V.dot(V.T)
    X = Y * np.outer(sc, sc)
    # The mixing iterate's <C, V V'> is an UPPER estimate of the SDP value (exact only at convergence); what is
    # published as sdr_bound is the rigorous dual bound  -sum(y) + N min(0, lambda_min(C + diag(y)))  of the
    # stationarity multipliers (valid for any V), and the solve is checked like a solver status.
    primal = float(hist[-1])
    y, lmin, lower = dual_certificate(C, V)
    info = dict(V=V, C=C, hist=hist, sweeps=sweeps, scale=sc, y=y, primal=primal, dual_bound=lower, infeas=0.0,
                sweep_limit_hit=bool(sweeps >= max_sweeps))
    certify(info, lmin, 1.0 + float(np.max(np.abs(C))), 'solve_sdr (mixing method)')
    return X, lower, info


# ------------------------------------------------------------------------- general QCQPs
def _functions(form, engine=None):
    if not hasattr(form, 'f0'):           # GeneratedForm: the functions only exist inside the context
        Q, r, rel = engine.linear_terms()
        return Q, r, np.array([x == '==' for x in rel])
    fs = [form.f0] + list(form.fs)
    Q = np.array([np.asarray(f.qarray, dtype=np.float64).ravel() for f in fs])      # (m+1, n)
    r = np.array([f.r for f in fs])
    eq = np.array([f.relop == '==' for f in fs])
    return Q, r, eq


class _LbfgsResult(object):
    def __init__(self, x, f, nit, nfev):
        self.x, self.fun, self.nit, self.nfev = x, f, nit, nfev


def _lbfgs(fun_grad, x0, maxiter, maxfun, gtol, ftol, memory=20):
    """Unconstrained L-BFGS (two-loop recursion, Armijo backtracking with a cubic / quadratic step model, curvature-guarded
    updates) for the inner problems of the augmented Lagrangian.  Same stopping rules as SciPy's L-BFGS-B -- relative decrease
    <= ftol, max |g| <= gtol, iteration / evaluation limits -- at a fraction of its host cost: on the 262 193 unknowns of the
    full-size factor one call of L-BFGS-B's Fortran driver takes 33 ms (bound handling, Cauchy point, subspace step), this
    loop 5 ms; the function / gradient evaluation it feeds is 70 ms of device work."""
    x = np.array(x0, dtype=np.float64)
    f, g = fun_grad(x)
    g = np.array(g, dtype=np.float64)
    nfev = 1
    S, Y, RHO = [], [], []
    nit = 0
    while nit < maxiter and nfev < maxfun:
        if np.max(np.abs(g)) <= gtol:
            break
        # two-loop recursion: d = -H g
        q = g.copy()
        al = [0.0] * len(S)
        for i in range(len(S) - 1, -1, -1):
            al[i] = RHO[i] * S[i].dot(q)
            q -= al[i] * Y[i]
        if S:
            q *= S[-1].dot(Y[-1]) / Y[-1].dot(Y[-1])
        for i in range(len(S)):
            be = RHO[i] * Y[i].dot(q)
            q += (al[i] - be) * S[i]
        d = -q
        gd = g.dot(d)
        if not gd < 0.0:                     # not a descent direction: restart from steepest descent
            S, Y, RHO = [], [], []
            d = -g
            gd = -g.dot(g)
        t = 1.0 if S else min(1.0, 1.0 / max(np.sqrt(-gd), 1e-300))
        f_new = g_new = None
        t_prev = f_prev = None
        ok = False
        for _ in range(30):
            if nfev >= maxfun:
                break
            x_new = x + t * d
            f_new, g_new = fun_grad(x_new)
            nfev += 1
            if np.isfinite(f_new) and f_new <= f + 1e-4 * t * gd:
                ok = True
                break
            # minimiser of the quadratic through f, gd, f(t) (cubic with the previous trial when there is one), safeguarded
            if not np.isfinite(f_new):
                t_next = 0.1 * t
            elif t_prev is None:
                t_next = -gd * t * t / (2.0 * (f_new - f - gd * t))
            else:
                r1 = f_new - f - gd * t
                r2 = f_prev - f - gd * t_prev
                den = t * t * t_prev * t_prev * (t - t_prev)
                a_ = (t_prev * t_prev * r1 - t * t * r2) / den
                b_ = (-t_prev ** 3 * r1 + t ** 3 * r2) / den
                disc = b_ * b_ - 3.0 * a_ * gd
                t_next = (-b_ + np.sqrt(disc)) / (3.0 * a_) if (a_ != 0.0 and disc >= 0.0) else -gd / (2.0 * b_)
            t_prev, f_prev = t, f_new
            t = float(min(max(t_next, 0.1 * t), 0.5 * t)) if np.isfinite(t_next) else 0.1 * t
        if not ok:
            break
        g_new = np.array(g_new, dtype=np.float64)
        s = x_new - x
        y = g_new - g
        sy = s.dot(y)
        if sy > 1e-10 * np.sqrt(s.dot(s) * y.dot(y)):
            S.append(s); Y.append(y); RHO.append(1.0 / sy)
            if len(S) > memory:
                S.pop(0); Y.pop(0); RHO.pop(0)
        decrease = f - f_new
        x, g = x_new, g_new
        f_old, f = f, f_new
        nit += 1
        if decrease <= ftol * max(abs(f_old), abs(f), 1.0):
            break
    return _LbfgsResult(x, f, nit, nfev)


def _burer_monteiro_al(n, m, eq, sc, rank, values, gradient, sigma0, outer, inner, feas_tol, seed, verbose, optimizer='own',
                       inner0=None, growth=1.5, ftol_final=1e-15):
    """Augmented-Lagrangian loop on the Burer-Monteiro factor V ((n+1) x rank) of the lifted matrix, shared by the
    dense and the separable operator sets.  values(V) -> h (m+1,): <M_k, V V'> for the objective (k = 0) and every
    constraint; gradient(V, ws, wN) -> d/dV of sum_k ws_k <M_k, V V'> + wN |t|^2 (V is the matrix `values` saw last)."""
    from scipy.optimize import minimize
    ineq = ~eq
    ineq[0] = False
    eqc = eq.copy()
    eqc[0] = False
    rs = np.random.RandomState(seed)
    V = 0.1 * rs.randn(n + 1, rank)
    V[n, :] = 0.0
    V[n, 0] = 1.0
    y = np.zeros(m + 1)      # y[0] unused
    yN = 0.0
    sigma = float(sigma0)

    def fun_grad(v):
        Vm = v.reshape(n + 1, rank)
        t = Vm[n, :]
        h = values(Vm) * sc
        tt = t.dot(t) - 1.0
        w = np.zeros(m + 1)
        w[0] = 1.0
        w[eqc] = y[eqc] + sigma * h[eqc]
        w[ineq] = np.maximum(0.0, y[ineq] + sigma * h[ineq])
        wN = yN + sigma * tt
        L = h[0] + np.sum(y[eqc] * h[eqc] + 0.5 * sigma * h[eqc] ** 2) \
            + np.sum((w[ineq] ** 2 - y[ineq] ** 2) / (2.0 * sigma)) + yN * tt + 0.5 * sigma * tt * tt
        G = gradient(Vm, w * sc, wN)
        return L, G.ravel()

    hist = []
    prev_infeas = None
    infeas = np.inf
    for it in range(outer):
        # inexact inner solves: no point in polishing the Lagrangian far below the current infeasibility
        rough = prev_infeas is not None and prev_infeas > 1e-4
        ftol = 1e-9 if (prev_infeas is None or rough) else ftol_final
        # inexact inner solves, second part: while the multipliers are far from their limit a few dozen L-BFGS iterations per
        # outer iteration move them as well as three hundred do (inner0 iterations at first, growing by `growth` per outer
        # iteration up to `inner`)
        cap = inner if inner0 is None else int(min(inner, inner0 * growth ** it))
        if optimizer == 'scipy':
            res = minimize(fun_grad, V.ravel(), jac=True, method='L-BFGS-B',
                           options=dict(maxiter=cap, maxfun=2 * cap, gtol=1e-9, ftol=ftol, maxcor=20))
        else:
            res = _lbfgs(fun_grad, V.ravel(), cap, 2 * cap, 1e-9, ftol, memory=20)
        V = res.x.reshape(n + 1, rank)
        h = values(V) * sc
        tt = V[n, :].dot(V[n, :]) - 1.0
        infeas = max(np.max(np.abs(h[eqc])) if eqc.any() else 0.0, np.max(np.maximum(h[ineq], 0.0)) if ineq.any() else 0.0, abs(tt))
        y[eqc] = y[eqc] + sigma * h[eqc]
        y[ineq] = np.maximum(0.0, y[ineq] + sigma * h[ineq])
        yN = yN + sigma * tt
        hist.append((float(h[0]), float(infeas), sigma, int(res.nit)))
        prev_infeas = infeas
        if verbose:
            print('outer %2d: <C,X> %.8g infeas %.2e sigma %.1e inner its %d' % (it, h[0], infeas, sigma, res.nit))
        if infeas < feas_tol and it > 0 and abs(hist[-1][0] - hist[-2][0]) <= 1e-7 * (1.0 + abs(h[0])):
            break
        if it > 0 and infeas > 0.25 * hist[-2][1]:
            sigma = min(sigma * 5.0, 1e8)
    X = V.dot(V.T)
    X = X / X[n, n]
    bound = float(values(V)[0] / V[n, :].dot(V[n, :]))
    return X, bound, dict(V=V, y=y * sc, yN=yN, hist=hist, rank=rank, dual_value=-yN, primal=bound,
                          infeas=float(infeas), outer_limit_hit=(len(hist) >= outer and not infeas < feas_tol))


def _default_rank(m, rank):
    if rank is None:
        rank = int(np.ceil(np.sqrt(2.0 * (m + 2)))) + 1
    return int(min(max(rank, 2), 64))


def solve_sdr_general(engine, form, rank=None, sigma0=10.0, outer=25, inner=400, feas_tol=1e-6, seed=0, verbose=False, optimizer='own',
                      inner0=20, growth=1.3, ftol_final=1e-15):
    """SDP relaxation of ANY QCQP the dense path holds (constraints that couple coordinates), in the
    Burer-Monteiro form X = V V' (V: (n+1) x r, r(r+1)/2 > m+1) with an augmented Lagrangian on the
    constraints  <M_k, X> (<=, ==) 0,  X_nn = 1   (solve_sdr, qcqp.py:72-97).  The heavy linear algebra runs
    on the device through the engine: the constraint values are quadratic forms of the r columns of V
    (one batched evaluation with the quadratic and the linear parts of the homogeneous forms kept apart) and the
    gradient is 2 S V with S = sum_k w_k M_k (qcqpmi_pop_weighted_product: one pass over all matrices + one GEMM).
    The host runs L-BFGS on the (n+1) r entries of V and the multiplier updates.
    Returns (X, bound, info); info['y'] are the multipliers (info['dual_value'] = -y_N is a lower bound of the
    SDP value whenever C + sum y_k M_k + y_N E_NN is PSD -- certify_general checks that)."""
    n, m = form.n, form.m
    Q, rr, eq = _functions(form, engine)
    rank = _default_rank(m, rank)
    # row scaling of the constraints (the multipliers are un-scaled at the end)
    sc = np.ones(m + 1)
    sc[1:] = 1.0 / (1.0 + np.abs(rr[1:]) + np.linalg.norm(Q[1:], axis=1) / np.sqrt(n))
    evals = [0]
    import time as _time
    tm = dict(upload=0.0, eval_parts=0.0, weighted_product=0.0, host_values=0.0, host_gradient=0.0)

    def values(Vm):
        Vx, t = Vm[:n, :], Vm[n, :]
        t0 = _time.perf_counter()
        engine.upload(Vx)
        t1 = _time.perf_counter()
        quad, lin = engine.eval_parts()          # x'P_k x + r_k and q_k'x per column, one pass on the device
        t2 = _time.perf_counter()
        evals[0] += 1
        h = (quad - rr[:, None]).sum(axis=1) + lin.dot(t) + rr * t.dot(t)
        tm['upload'] += t1 - t0; tm['eval_parts'] += t2 - t1; tm['host_values'] += _time.perf_counter() - t2
        return h

    def gradient(Vm, ws, wN):
        Vx, t = Vm[:n, :], Vm[n, :]
        t0 = _time.perf_counter()
        SVx = engine.weighted_product(ws)                       # population = Vx (uploaded by values()): (sum_k ws_k P_k) Vx
        t1 = _time.perf_counter()
        tm['weighted_product'] += t1 - t0
        try:
            return _gradient_host(Vx, t, SVx, ws, wN)
        finally:
            tm['host_gradient'] += _time.perf_counter() - t1

    def _gradient_host(Vx, t, SVx, ws, wN):
        qh = 0.5 * ws.dot(Q)                                    # sum_k ws_k q_k / 2
        G = np.empty((n + 1, Vx.shape[1]))
        G[:n, :] = 2.0 * (SVx + np.outer(qh, t))
        G[n, :] = 2.0 * (qh.dot(Vx) + (ws.dot(rr) + wN) * t)
        return G

    t_all = _time.perf_counter()
    # OpenBLAS starts one spinning thread per core (64 on the GPU box) for the small products of the optimizer: they starve
    # the HIP runtime's own threads -- 25 ms per device call at full size -- and make the level-1 products slower, not faster
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=4, user_api='blas')
    except Exception:       # noqa: BLE001 -- optional dependency
        limiter = None
    try:
        X, bound, info = _burer_monteiro_al(n, m, eq, sc, rank, values, gradient, sigma0, outer, inner, feas_tol, seed, verbose, optimizer,
                                            inner0, growth, ftol_final)
    finally:
        if limiter is not None:
            limiter.restore_original_limits()
    info['evals'] = evals[0]
    tm['total'] = _time.perf_counter() - t_all
    tm['optimizer_and_rest'] = tm['total'] - sum(v for k, v in tm.items() if k != 'total')
    info['timing'] = tm          # seconds: where the wall time of the solve went
    return X, bound, info


def separable_family(form):
    """(coord, p, q, r) arrays (one entry per constraint) if every constraint of the form touches exactly one
    coordinate -- boxes, discs / annuli in one variable, one-sided bounds, x_i^2 = d_i -- else None."""
    if not hasattr(form, 'fs'):
        return None
    m = form.m
    coord = np.zeros(m, dtype=np.int64); p = np.zeros(m); q = np.zeros(m); r = np.zeros(m)
    for k, f in enumerate(form.fs):
        qa = np.asarray(f.qarray, dtype=np.float64).ravel()
        if hasattr(f.P, 'tocoo'):
            Pc = f.P.tocoo()
            keep = Pc.data != 0.0
            rows, cols, vals = Pc.row[keep], Pc.col[keep], Pc.data[keep]
        else:
            P = np.asarray(f.P)
            rows, cols = np.nonzero(P)
            vals = P[rows, cols]
        qi = np.nonzero(qa)[0]
        idx = set(int(a) for a in rows) | set(int(a) for a in cols) | set(int(a) for a in qi)
        if len(idx) != 1:
            return None
        i = idx.pop()
        coord[k] = i
        p[k] = float(vals.sum()) if len(vals) else 0.0
        q[k] = qa[i]
        r[k] = f.r
    return coord, p, q, r


def solve_sdr_separable(engine, form, rank=None, sigma0=10.0, outer=40, inner=400, feas_tol=1e-6, seed=0, verbose=False):
    """The same relaxation for problems whose constraints each touch ONE coordinate (any mix of boxes, discs,
    annuli, bounds, x_i^2 = d_i): <M_k, X> = p_k X_ii + q_k X_in + r_k X_nn is elementwise on the factor, only the
    objective's P0 V is a matrix product -- on the device (the engine's packed P0: qcqpmi_pop_eval for the quadratic
    form of every column, qcqpmi_pop_weighted_product for P0 V)."""
    fam = separable_family(form)
    if fam is None:
        return None
    coord, p, q, r = fam
    n, m = form.n, form.m
    q0 = np.asarray(form.f0.qarray, dtype=np.float64).ravel()
    r0 = float(form.f0.r)
    eq = np.array([False] + [f.relop == '==' for f in form.fs])
    rank = _default_rank(m, rank)
    sc = np.ones(m + 1)
    sc[1:] = 1.0 / (1.0 + np.abs(r) + np.abs(q) + np.abs(p))
    w0 = np.zeros(m + 1)
    w0[0] = 1.0

    def values(Vm):
        Vx, t = Vm[:n, :], Vm[n, :]
        engine.upload(Vx)
        f0, _ = engine.eval()                                    # x'P0x + q0'x + r0 per column
        tt = t.dot(t)
        h = np.empty(m + 1)
        h[0] = (f0 - Vx.T.dot(q0) - r0).sum() + q0.dot(Vx.dot(t)) + r0 * tt
        Xii = np.einsum('ik,ik->i', Vx, Vx)
        Xin = Vx.dot(t)
        h[1:] = p * Xii[coord] + q * Xin[coord] + r * tt
        return h

    def gradient(Vm, ws, wN):
        Vx, t = Vm[:n, :], Vm[n, :]
        P0V = engine.weighted_product(w0)                        # P0 Vx on the device (population = Vx)
        pw = np.bincount(coord, weights=ws[1:] * p, minlength=n)     # sum of w_k p_k on each coordinate
        qw = np.bincount(coord, weights=ws[1:] * q, minlength=n)
        qh = 0.5 * (ws[0] * q0 + qw)
        G = np.empty_like(Vm)
        G[:n, :] = 2.0 * (ws[0] * P0V + pw[:, None] * Vx + np.outer(qh, t))
        G[n, :] = 2.0 * (qh.dot(Vx) + (ws[0] * r0 + ws[1:].dot(r) + wN) * t)
        return G

    X, bound, info = _burer_monteiro_al(n, m, eq, sc, rank, values, gradient, sigma0, outer, inner, feas_tol, seed, verbose)
    info['family'] = 'separable'
    return X, bound, info


def certify(info, lmin, scale, what):
    """Turn the raw outcome of a relaxation solve into what may be published as a BOUND (the reference only accepts
    solver status OPTIMAL / OPTIMAL_INACCURATE, qcqp.py:94-95).  info['converged'] is True when the iterate is feasible
    to tolerance, the iteration limit was not the reason to stop and the dual slack matrix is PSD to rounding
    (lambda_min >= -1e-6 scale).  Raises when the solve is clearly not a solution of the relaxation."""
    import logging
    log = logging.getLogger('qcqp_amd')
    ok_psd = lmin >= -1e-6 * scale
    ok_feas = info.get('infeas', 0.0) <= 1e-5
    limit = bool(info.get('outer_limit_hit', False)) or bool(info.get('sweep_limit_hit', False))
    info['lambda_min'] = float(lmin)
    info['converged'] = bool(ok_psd and ok_feas and not limit)
    if not info['converged']:
        msg = ('%s: relaxation not solved to optimality (lambda_min of the dual slack %.3e, infeasibility %.2e, iteration '
               'limit hit: %s)' % (what, lmin, info.get('infeas', 0.0), limit))
        if lmin < -1e-2 * scale or info.get('infeas', 0.0) > 1e-2:
            raise Exception("Relaxation problem status: " + msg)
        log.warning(msg + '; the published bound is the rigorous / dual value where one exists')
    return info


def dual_slack_separable(form, fam, y, yN):
    """S = M0 + sum_k y_k M_k + y_N E_NN for separable constraints (host, O(n^2)): the constraints only touch the
    diagonal and the last row / column of the lifted matrix."""
    coord, p, q, r = fam
    n = form.n
    P0 = _dense(form.f0.P)
    S = np.zeros((n + 1, n + 1))
    S[:n, :n] = 0.5 * (P0 + P0.T)
    yk = y[1:]
    S[np.arange(n), np.arange(n)] += np.bincount(coord, weights=yk * p, minlength=n)
    qh = 0.5 * (np.asarray(form.f0.qarray, dtype=np.float64).ravel() + np.bincount(coord, weights=yk * q, minlength=n))
    S[:n, n] = qh
    S[n, :n] = qh
    S[n, n] = form.f0.r + yk.dot(r) + yN
    return float(np.linalg.eigvalsh(S)[0]), S


def dual_certificate_device(engine, y, yN):
    """The same check when the matrices only exist on the device: S_P = sum w_k P_k comes from the engine."""
    Q, rr, _rel = engine.linear_terms()
    n = engine.n
    w = np.append(1.0, y[1:])
    S = np.zeros((n + 1, n + 1))
    SP = engine.weighted_matrix(w)
    S[:n, :n] = 0.5 * (SP + SP.T)
    qh = 0.5 * w.dot(Q)
    S[:n, n] = qh; S[n, :n] = qh
    S[n, n] = w.dot(rr) + yN
    return float(np.linalg.eigvalsh(S)[0]), S


def dual_certificate_general(form, y, yN):
    """lambda_min of S = C + sum_k y_k M_k + y_N E_NN (homogeneous forms, utilities.py:66-67) on the host;
    with S PSD and y_k >= 0 on the inequalities, -y_N is a lower bound of the SDP (hence of the QCQP)."""
    n = form.n
    fs = [form.f0] + list(form.fs)
    S = np.zeros((n + 1, n + 1))
    w = np.append(1.0, y[1:])
    for wk, f in zip(w, fs):
        if wk == 0.0:
            continue
        P = _dense(f.P)
        q = np.asarray(f.qarray, dtype=np.float64).ravel()
        S[:n, :n] += wk * 0.5 * (P + P.T)
        S[:n, n] += 0.5 * wk * q
        S[n, :n] += 0.5 * wk * q
        S[n, n] += wk * f.r
    S[n, n] += yN
    return float(np.linalg.eigvalsh(S)[0]), S


# ------------------------------------------------------------------------- spectral relaxation
def solve_spectral(form, device=0, seed=0):
    """solve_spectral (qcqp.py:41-70): the relaxation with ALL inequality constraints summed into one and all
    equality constraints summed into one,
        minimise <W0, X>  s.t.  <W1, X> <= 0,  <W2, X> == 0,  X_nn = 1,  X PSD,
    solved by the general solver above on an auxiliary three-function problem; returns
    (sqrt(lambda_max) * v[:-1], value) exactly like the reference builds its point from the top eigenpair of X."""
    from .engine import Engine
    from .form import QCQPForm
    n = form.n
    P0 = _dense(form.f0.P)
    funcs = [(0.5 * (P0 + P0.T), np.asarray(form.f0.qarray, dtype=np.float64).ravel(), form.f0.r, None)]
    for relop in ('<=', '=='):
        P = np.zeros((n, n)); q = np.zeros(n); r = 0.0
        any_ = False
        for f in form.fs:
            if f.relop != relop:
                continue
            any_ = True
            Pk = _dense(f.P)
            P += 0.5 * (Pk + Pk.T); q += np.asarray(f.qarray, dtype=np.float64).ravel(); r += f.r
        if any_:
            funcs.append((P, q, r, relop))
    aux = QCQPForm.from_arrays(funcs)
    e = Engine(aux, device=device)
    try:
        if e.separable:
            raise Exception("spectral relaxation: the aggregated constraints touch a single coordinate "
                            "(one-variable problem); nothing to relax")
        e.L.qcqpmi_debug_profile(e.h, 32 << 4, None)      # the dense path also for n <= 64
        X, value, info = solve_sdr_general(e, aux, seed=seed)
    finally:
        e.close()
    w, v = np.linalg.eigh(X)
    x = np.sqrt(max(w[-1], 0.0)) * v[:-1, -1]
    return x, value, dict(X=X, sdr=info, aux=aux)
