"""Synthetic QCQP generators (host side, NumPy only).

They mirror the problem families of the reference's examples and of
BASELINE.json's configs; each returns a list of ``(P, q, r, relop)`` tuples
with the objective first (relop ``None``) -- the raw-array form accepted by
``qcqp_amd.QCQPForm.from_arrays`` -- plus a ``maximize`` flag.

* ``boolean_least_squares``  -- examples/boolean_least_squares.py:6-15
* ``maxcut``                 -- examples/maxcut.py:9-21
* ``box_least_squares``      -- the box-constrained sibling of the Boolean family (one interval per coordinate)
* ``beamforming``            -- examples/secondary_user_beamforming.py:18-41
* ``dense_indefinite``       -- SURVEY.md section 8(d) cfg5 generator
* ``circle_packing``         -- examples/circle_packing.py:6-17 (two variables: centres 2 x N and the radius)
"""
import numpy as np
import scipy.sparse as sp


def boolean_least_squares(n, m_rows, seed=1, legacy_seed=False):
    """minimize ||Ax-b||^2  s.t. x_i^2 == 1.

    legacy_seed=True draws A, b exactly like the example script
    (np.random.seed(seed); randn(m, n); randn(m, 1))."""
    if legacy_seed:
        st = np.random.get_state()
        np.random.seed(seed)
        A = np.random.randn(m_rows, n)
        b = np.random.randn(m_rows, 1)
        np.random.set_state(st)
    else:
        rs = np.random.RandomState(seed)
        A = rs.randn(m_rows, n)
        b = rs.randn(m_rows, 1)
    P0 = A.T.dot(A)
    P0 = (P0 + P0.T) / 2.
    q0 = (-2. * A.T.dot(b)).ravel()
    r0 = float(b.T.dot(b)[0, 0])
    funcs = [(P0, q0, r0, None)]
    for i in range(n):
        P = sp.csr_matrix(([1.0], ([i], [i])), shape=(n, n))
        funcs.append((P, np.zeros(n), -1.0, '=='))
    return funcs, False, dict(A=A, b=b)


def box_least_squares(n, m_rows, bound=1.0, seed=1, ridge=0.1):
    """minimize ||Ax-b||^2 + ridge ||x||^2  s.t. x_i^2 <= bound^2  (the box -bound <= x_i <= bound as one convex quadratic per
    coordinate: ONE constraint class, one interval per coordinate -- the relaxed sibling of boolean_least_squares)."""
    rs = np.random.RandomState(seed)
    A = rs.randn(m_rows, n)
    b = rs.randn(m_rows, 1) * np.sqrt(n)
    P0 = A.T.dot(A) + ridge * np.eye(n)
    P0 = (P0 + P0.T) / 2.
    q0 = (-2. * A.T.dot(b)).ravel()
    r0 = float(b.T.dot(b)[0, 0])
    funcs = [(P0, q0, r0, None)]
    for i in range(n):
        P = sp.csr_matrix(([1.0], ([i], [i])), shape=(n, n))
        funcs.append((P, np.zeros(n), -float(bound) ** 2, '<='))
    return funcs, False, dict(A=A, b=b)


def multi_class(name, n, seed=1):
    """Separable problems with SEVERAL constraint classes and up to two constraints per coordinate (what cd_life_kernel's GENK / LINK
    kinds take; the reference treats every coordinate's list on its own, qcqp.py:113-141, 160-176).  Returns funcs (objective first).
      box3   least squares + ridge; coordinate i carries x^2 <= 1 | x^2 <= 0.49 | x^2 - x/2 - 1/2 <= 0 (= [-1/2, 1]) by i mod 3
      ann2   least squares + ridge; even i: x^2 <= 1 AND -x^2 <= -1/4 (two intervals [-1, -1/2] u [1/2, 1]); odd i: x^2 == 1
      lin2   least squares + ridge; TWO LINEAR constraints per coordinate, x <= u_c and -x <= -l_c, bounds by i mod 2
      cut2   weighted MAXCUT objective (zero diagonal); even i: x^2 == 1, odd i: x^2 <= 1 (a relaxed vertex)"""
    rs = np.random.RandomState(seed)
    funcs = []
    if name == 'cut2':
        U = np.triu((rs.uniform(size=(n, n)) < 0.5).astype(float), 1) * np.triu(rs.uniform(0.5, 1.5, size=(n, n)), 1)
        W = U + U.T
        funcs.append((0.25 * W, np.zeros(n), -0.25 * float(W.sum()), None))
    else:
        m_rows = max(4, n // 2)
        A = rs.randn(m_rows, n)
        b = rs.randn(m_rows, 1) * np.sqrt(n)
        P0 = A.T.dot(A) + 0.1 * np.eye(n)
        funcs.append(((P0 + P0.T) / 2., (-2. * A.T.dot(b)).ravel(), float(b.T.dot(b)[0, 0]), None))

    def quad(i, p, q, r, relop):
        P = sp.csr_matrix(([float(p)], ([i], [i])), shape=(n, n))
        qv = np.zeros(n)
        qv[i] = float(q)
        funcs.append((P, qv, float(r), relop))
    for i in range(n):
        if name == 'box3':
            if i % 3 == 0:
                quad(i, 1.0, 0.0, -1.0, '<=')
            elif i % 3 == 1:
                quad(i, 1.0, 0.0, -0.49, '<=')
            else:
                quad(i, 1.0, -0.5, -0.5, '<=')
        elif name == 'ann2':
            if i % 2 == 0:
                quad(i, 1.0, 0.0, -1.0, '<=')
                quad(i, -1.0, 0.0, 0.25, '<=')
            else:
                quad(i, 1.0, 0.0, -1.0, '==')
        elif name == 'lin2':
            lo, hi = ((-0.75, 1.25), (-1.5, 0.5))[i % 2]
            quad(i, 0.0, 1.0, -hi, '<=')
            quad(i, 0.0, -1.0, lo, '<=')
        elif name == 'cut2':
            quad(i, 1.0, 0.0, -1.0, '==' if i % 2 == 0 else '<=')
        else:
            raise KeyError(name)
    return funcs


def maxcut(n, p=0.5, seed=1, weighted=False):
    """maximize 0.25 (sum(W) - x^T W x)  s.t. x_i^2 == 1   (minimise form returned).
    weighted=True draws edge weights from U(0.5, 1.5): no exact ties between cuts."""
    rs = np.random.RandomState(seed)
    U = np.triu((rs.uniform(size=(n, n)) < p).astype(float), 1)
    if weighted:
        U = U * np.triu(rs.uniform(0.5, 1.5, size=(n, n)), 1)
    W = U + U.T
    P0 = 0.25 * W
    r0 = -0.25 * float(W.sum())
    funcs = [(P0, np.zeros(n), r0, None)]
    for i in range(n):
        P = sp.csr_matrix(([1.0], ([i], [i])), shape=(n, n))
        funcs.append((P, np.zeros(n), -1.0, '=='))
    return funcs, True, dict(W=W)


def beamforming(nant, m_h, l, tau=20., eta=2., seed=1):
    """minimize ||x||^2 s.t. |h_i^H x|^2 >= tau, |g_i^H x|^2 <= eta, real expansion (n = 2 nant)."""
    rs = np.random.RandomState(seed)
    HR = rs.randn(m_h, nant)
    HI = rs.randn(m_h, nant)
    A = np.hstack((HR, HI))
    B = np.hstack((-HI, HR))
    GR = rs.randn(l, nant)
    GI = rs.randn(l, nant)
    Cm = np.hstack((GR, GI))
    D = np.hstack((-GI, GR))
    n = 2 * nant
    funcs = [(np.eye(n), np.zeros(n), 0.0, None)]
    for i in range(m_h):
        P = -(np.outer(A[i], A[i]) + np.outer(B[i], B[i]))
        funcs.append((P, np.zeros(n), tau, '<='))
    for i in range(l):
        P = np.outer(Cm[i], Cm[i]) + np.outer(D[i], D[i])
        funcs.append((P, np.zeros(n), -eta, '<='))
    return funcs, False, dict(A=A, B=B, C=Cm, D=D)


def dense_indefinite(n, m, seed=7, easy=True):
    """Random indefinite dense QCQP: P_k = (G+G^T)/2, G = randn/sqrt(n); last constraint is the
    ball ||x||^2 <= n.  easy=True scales r_k with n so that feasibility is reachable."""
    rs = np.random.RandomState(seed)

    def sym():
        G = rs.randn(n, n) / np.sqrt(n)
        return (G + G.T) / 2.

    funcs = [(sym(), rs.randn(n), 0.0, None)]
    for k in range(m - 1):
        r = -1. - abs(rs.randn())
        if easy:
            r *= n / 8.
        funcs.append((sym(), rs.randn(n), r, '<='))
    funcs.append((np.eye(n), np.zeros(n), -float(n), '<='))
    return funcs, False, {}


def circle_packing(N, B=10., minimize_form=False):
    """maximize r  s.t.  r <= X <= B - r, r >= 0, (2r)^2 <= ||X[:, i] - X[:, j]||^2 for i < j
    (examples/circle_packing.py:6-17: N circles of radius r in the box [0, B]^2).  Two cvxpy variables in the reference --
    X (2, N) and the scalar r; get_qcqp_form (utilities.py:318-347) stacks them in the order prob.variables() gives them,
    each flattened COLUMN-major (utilities.py:298-316): here x = [X[0,0], X[1,0], X[0,1], ..., X[1,N-1], r], n = 2 N + 1,
    var_sizes [(2, N), (1, 1)].  Constraints in the order of the example's list: 2 N of r - X[d, i] <= 0 (column-major),
    2 N of X[d, i] + r - B <= 0, -r <= 0, then the N (N - 1) / 2 separation constraints 4 r^2 - ||X_i - X_j||^2 <= 0 (sparse,
    indefinite, five coordinates each).  Returns (funcs, maximize, info): funcs as stated (objective = r, to be maximised;
    minimize_form=True: already negated, what QCQPForm holds), info['var_sizes']."""
    import scipy.sparse as sp
    n = 2 * N + 1
    ir = 2 * N
    q0 = np.zeros(n)
    q0[ir] = -1.0 if minimize_form else 1.0
    funcs = [(sp.csr_matrix((n, n)), q0, 0.0, None)]
    Z = sp.csr_matrix((n, n))
    for j in range(2 * N):                       # X >= r
        q = np.zeros(n)
        q[ir], q[j] = 1.0, -1.0
        funcs.append((Z, q, 0.0, '<='))
    for j in range(2 * N):                       # X <= B - r
        q = np.zeros(n)
        q[ir], q[j] = 1.0, 1.0
        funcs.append((Z, q, -float(B), '<='))
    q = np.zeros(n)
    q[ir] = -1.0
    funcs.append((Z, q, 0.0, '<='))              # r >= 0
    for i in range(N):
        for j in range(i + 1, N):
            rows, cols, vals = [ir], [ir], [4.0]
            for d in range(2):
                a, b = 2 * i + d, 2 * j + d
                rows += [a, b, a, b]
                cols += [a, b, b, a]
                vals += [-1.0, -1.0, 1.0, 1.0]
            funcs.append((sp.csr_matrix((vals, (rows, cols)), shape=(n, n)), np.zeros(n), 0.0, '<='))
    return funcs, (not minimize_form), dict(var_sizes=[(2, N), (1, 1)], B=float(B))


class GeneratedForm(object):
    """A QCQP whose dense functions are synthesised ON THE DEVICE from a seed
    (qcqpmi_set_quad_generated; BASELINE.json configs[4] at full size is 137.6 GB of matrices).
    specs[0] is the objective; every entry: seed, scale, qscale, diag_add, r, relop."""

    def __init__(self, n, specs):
        self.n, self.m, self.specs = int(n), len(specs) - 1, list(specs)


def dense_indefinite_generated(n, m, seed=7, easy=True):
    """The dense_indefinite family with device-side generation: P_k ~ (G+G^T)/2, G = randn/sqrt(n)
    (same law, symmetric by construction), q_k = randn, r_k = -(1+|z_k|) [* n/8], last constraint the
    ball ||x||^2 <= n."""
    rs = np.random.RandomState(seed)
    sc = 1.0 / np.sqrt(n)
    specs = [dict(seed=seed, scale=sc, qscale=1.0, diag_add=0.0, r=0.0, relop=None)]
    for k in range(m - 1):
        r = -1. - abs(rs.randn())
        if easy:
            r *= n / 8.
        specs.append(dict(seed=seed, scale=sc, qscale=1.0, diag_add=0.0, r=r, relop='<='))
    specs.append(dict(seed=seed, scale=0.0, qscale=0.0, diag_add=1.0, r=-float(n), relop='<='))
    return GeneratedForm(n, specs)


def materialise_generated(form, keyed_normal):
    """The same functions as NumPy arrays (small n only): keyed_normal(seed, stream, elem) is the
    oracle's orc_keyed_normal.  Returns the usual [(P, q, r, relop), ...] list."""
    n = form.n
    funcs = []
    for k, g in enumerate(form.specs):
        P = np.zeros((n, n))
        q = np.zeros(n)
        for i in range(n):
            for j in range(i, n):
                v = g['scale'] * keyed_normal(g['seed'], (1 << 48) + k, i * n + j) if g['scale'] != 0.0 else 0.0
                if i == j:
                    P[i, i] = v + g['diag_add']
                else:
                    P[i, j] = P[j, i] = v * 0.70710678118654752440
            if g['qscale'] != 0.0:
                q[i] = g['qscale'] * keyed_normal(g['seed'], (2 << 48) + k, i)
        funcs.append((P, q, g['r'], g['relop']))
    return funcs
