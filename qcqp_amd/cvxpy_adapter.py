"""Front end for cvxpy problems: the job of the reference's ``get_qcqp_form`` (utilities.py:318-347), which
needs cvxpy 0.4's ``QuadCoeffExtractor``.  This adapter only relies on what every cvxpy >= 1.0 problem (and any
duck-typed stand-in) offers publicly: ``problem.objective`` (``NAME``, ``args[0]`` / ``expr``),
``problem.constraints`` (each with ``args`` or ``expr``), ``problem.variables()`` (``shape`` / ``size``, ``value``),
``expression.value`` and, when present, ``is_quadratic()``.

The coefficients are read off by EVALUATION: a quadratic f(x) = x'Px + q'x + r is determined by its values at
0, +-e_i and e_i + e_j (exact for a quadratic up to rounding -- these are not finite-difference approximations):
    r = f(0),  q_i = (f(e_i) - f(-e_i)) / 2,  P_ii = (f(e_i) + f(-e_i)) / 2 - r,
    P_ij = P_ji = (f(e_i + e_j) - f(e_i) - f(e_j) + r) / 2.
1 + 2n + 2(n+1) + (pairs of variables that share a function) <= 3 + 4n + n(n-1)/2 evaluations of all expressions at once: meant for the problem sizes one writes by hand in
cvxpy (n up to a few hundred); large instances should be passed as raw arrays (qcqp_amd.Problem).
One QuadraticFunction per SCALAR entry of every constraint, like utilities.py:341-345; P is symmetric by
construction (utilities.py:333, 345); a maximised objective is negated (utilities.py:335-336).
"""
import numpy as np

from .form import QCQPForm


def _nelem(v):
    shape = getattr(v, 'shape', None)
    if shape is None:
        shape = v.size          # cvxpy 0.4 style (rows, cols)
    if isinstance(shape, int):
        return int(shape)
    k = 1
    for a in shape:
        k *= int(a)
    return k


def _shape2(v):
    shape = getattr(v, 'shape', None)
    if shape is None:
        shape = v.size
    if isinstance(shape, int):
        return (int(shape), 1)
    shape = tuple(int(a) for a in shape)
    if len(shape) == 0:
        return (1, 1)
    if len(shape) == 1:
        return (shape[0], 1)
    return (shape[0], int(np.prod(shape[1:])))


class _VarAdapter(object):
    """What QCQP.assign_vars / flatten_vars expect (``size`` = (rows, cols), ``value``, ``id``), forwarding to the
    cvxpy variable with its own shape."""

    def __init__(self, var):
        self._v = var
        self.size = _shape2(var)
        self.id = getattr(var, 'id', id(var))

    @property
    def value(self):
        val = self._v.value
        if val is None:
            return None
        return np.reshape(np.asarray(val, dtype=np.float64), self.size, order='F')

    @value.setter
    def value(self, val):
        shape = getattr(self._v, 'shape', None)
        if val is None:
            self._v.value = None
            return
        arr = np.asarray(val, dtype=np.float64)
        if isinstance(shape, tuple):
            arr = np.reshape(arr, shape, order='F') if len(shape) else float(arr.ravel()[0])
        self._v.value = arr


class _Objective(object):
    def __init__(self, name):
        self.NAME = name


class CvxpyProblem(object):
    """Stand-in with the three things QCQP uses: qcqp_form, objective.NAME, variables()."""

    def __init__(self, form, name, variables):
        self.qcqp_form = form
        self.objective = _Objective(name)
        self._vars = variables

    def variables(self):
        return self._vars


def _expr_of(c):
    e = getattr(c, 'expr', None)
    if e is not None:
        return e
    args = c.args
    return args[0] - args[1] if len(args) == 2 else args[0]


def _relation(c):
    """(relop, sign): the constraint reads  sign * expr  relop  0."""
    name = type(c).__name__
    opn = getattr(c, 'OP_NAME', None)
    if opn == '==' or name in ('Equality', 'Zero', 'EqConstraint'):
        return '==', 1.0
    if name in ('NonNeg',) or opn == '>=':
        return '<=', -1.0
    if opn == '<=' or name in ('Inequality', 'NonPos', 'LeqConstraint'):
        return '<=', 1.0
    raise Exception("Unsupported constraint type for a QCQP: %s" % name)


def problem_from_cvxpy(prob):
    """cvxpy (or duck-typed) problem -> CvxpyProblem holding the QCQPForm.  Raises the reference's messages when
    the objective or a constraint is not quadratic (utilities.py:322-325)."""
    obj = prob.objective
    oexpr = obj.args[0] if hasattr(obj, 'args') and len(obj.args) else obj.expr
    if hasattr(oexpr, 'is_quadratic') and not oexpr.is_quadratic():
        raise Exception("Objective is not quadratic.")
    cons = list(prob.constraints)
    cexprs = [_expr_of(c) for c in cons]
    for e in cexprs:
        if hasattr(e, 'is_quadratic') and not e.is_quadratic():
            raise Exception("Not all constraints are quadratic.")
    rels = [_relation(c) for c in cons]
    xs = list(prob.variables())
    sizes = [_nelem(v) for v in xs]
    n = int(sum(sizes))
    saved = [v.value for v in xs]
    adapters = [_VarAdapter(v) for v in xs]

    def evaluate(x):
        ind = 0
        for a, k in zip(adapters, sizes):
            a.value = x[ind:ind + k]
            ind += k
        vals = [np.atleast_1d(np.asarray(oexpr.value, dtype=np.float64)).ravel(order='F')]
        for e in cexprs:
            vals.append(np.atleast_1d(np.asarray(e.value, dtype=np.float64)).ravel(order='F'))
        return np.concatenate(vals)

    try:
        z = np.zeros(n)
        f00 = evaluate(z)
        K = f00.size                     # scalar functions: objective + every constraint entry
        fp = np.zeros((n, K))
        fm = np.zeros((n, K))
        for i in range(n):
            z[i] = 1.0
            fp[i] = evaluate(z)
            z[i] = -1.0
            fm[i] = evaluate(z)
            z[i] = 0.0
        q = 0.5 * (fp - fm)                                   # (n, K)
        P = np.zeros((K, n, n))
        P[:, np.arange(n), np.arange(n)] = (0.5 * (fp + fm) - f00[None, :]).T
        # Only pairs that can interact: function k has no P_ij unless it DEPENDS on both x_i and x_j.  "Depends" is tested
        # at generic points, not on the axes alone (a bilinear term x_i x_j leaves f unchanged along either axis: the
        # round-2 test `f(+-e_i) != f(0)` dropped such P_ij silently): f(v + e_i) != f(v) for two fixed pseudo-random v.
        # An expression that does not contain x_i evaluates bit-identically, so the test has no false negatives except on
        # a measure-zero set of v (two independent v make that set empty for practical purposes) and no tolerance.
        touched = (fp != f00[None, :]) | (fm != f00[None, :])  # (n, K)
        prng = np.random.RandomState(0x5eed)
        for _ in range(2):
            v = prng.uniform(0.5, 1.5, size=n) * prng.choice([-1.0, 1.0], size=n)
            fv = evaluate(v)
            for i in range(n):
                keep = v[i]
                v[i] = keep + 1.0
                touched[i] |= (evaluate(v) != fv)
                v[i] = keep
        for i in range(n):
            for j in range(i + 1, n):
                if not np.any(touched[i] & touched[j]):
                    continue
                z[i] = 1.0
                z[j] = 1.0
                fij = evaluate(z)
                z[i] = 0.0
                z[j] = 0.0
                pij = 0.5 * (fij - fp[i] - fp[j] + f00)
                P[:, i, j] = pij
                P[:, j, i] = pij
    finally:
        for v, val in zip(xs, saved):
            v.value = val
    name = 'maximize' if getattr(obj, 'NAME', 'minimize') == 'maximize' else 'minimize'
    sgn = -1.0 if name == 'maximize' else 1.0             # utilities.py:335-336
    funcs = [(sgn * P[0], sgn * q[:, 0], sgn * float(f00[0]), None)]
    k = 1
    for e, (relop, s) in zip(cexprs, rels):
        shape = getattr(e, 'shape', None)
        if shape is None or isinstance(shape, int):
            cnt = int(shape) if isinstance(shape, int) else 1
        else:
            cnt = int(np.prod(shape)) if len(shape) else 1
        for _ in range(cnt):
            funcs.append((s * P[k], s * q[:, k], s * float(f00[k]), relop))
            k += 1
    assert k == K, 'constraint shapes do not match their values'
    form = QCQPForm.from_arrays(funcs)
    return CvxpyProblem(form, name, adapters)
