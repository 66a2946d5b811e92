"""Duck-typed stand-ins for the public surface of a cvxpy >= 1 problem (cvxpy is not installed in the image): what
qcqp_amd.cvxpy_adapter reads -- objective.NAME / args, constraint.expr / args, variables() with shape / value / id,
expression.value / shape / is_quadratic().  Shared by the CPU test of the adapter and the GPU test that drives a problem
through the adapter into the HIP path."""


class Var(object):
    def __init__(self, shape, vid=7):
        self.shape, self.value, self.id = shape, None, vid


class Expr(object):
    def __init__(self, fn, shape, quad=True):
        self.fn, self.shape, self.quad = fn, shape, quad
    value = property(lambda self: self.fn())

    def is_quadratic(self):
        return self.quad


class Objective(object):
    def __init__(self, name, e):
        self.NAME, self.args = name, [e]


class Equality(object):
    def __init__(self, e):
        self.expr = e


class Inequality(object):
    def __init__(self, e):
        self.expr = e


class NonNeg(object):
    def __init__(self, e):
        self.args = [e]


class Prob(object):
    def __init__(self, o, cs, vs):
        self.objective, self.constraints, self._vs = o, cs, vs

    def variables(self):
        return self._vs
