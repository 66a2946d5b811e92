"""Host-side mirror of the reference's problem containers (no arithmetic lives here).

``QuadraticFunction`` / ``QCQPForm`` keep the reference's field names
(qcqp/utilities.py:41-46, 122-130) so that code written against the reference's
``qcqp.qcqp_form`` keeps working; all evaluation is delegated to the HIP engine.
"""
import numpy as np
import scipy.sparse as sp

RELOP_CODE = {None: 0, '<=': 1, '==': 2}


class QuadraticFunction(object):
    """x^T P x + q^T x + r with optional relop '<=' / '=='  (utilities.py:41-46)."""

    def __init__(self, P, q, r, relop=None):
        if relop not in RELOP_CODE:
            raise Exception("Unknown relation operator: %s" % relop)
        n = int(np.asarray(q if not sp.issparse(q) else q.todense()).size)
        if sp.issparse(P):
            P = sp.csr_matrix(P, dtype=np.float64)
        else:
            P = np.ascontiguousarray(np.asarray(P, dtype=np.float64))
        assert P.shape == (n, n)
        self.P = P
        self.q = q
        self.qarray = np.ascontiguousarray(
            np.asarray(q.todense() if sp.issparse(q) else q, dtype=np.float64).ravel())
        self.r = float(r)
        self.relop = relop
        self.eigh = None  # for ADMM (utilities.py:46)


class QCQPForm(object):
    """f0 + list of constraint functions (utilities.py:122-130)."""

    def __init__(self, f0, fs):
        assert all([f.relop is not None for f in fs])
        self.f0 = f0
        self.fs = fs
        self.n = f0.P.shape[0]
        self.m = len(fs)
        self.rho = None       # for ADMM
        self.z_solver = None  # for ADMM

    def fi(self, i):
        return self.fs[i]

    @classmethod
    def from_arrays(cls, funcs):
        """funcs = [(P, q, r, relop), ...] with the objective first (relop None).
        P is symmetrised like get_qcqp_form does (utilities.py:333, 345)."""
        qs = []
        for (P, q, r, relop) in funcs:
            if sp.issparse(P):
                P = sp.csr_matrix((P + P.T) / 2.)
            else:
                P = np.asarray(P, dtype=np.float64)
                P = (P + P.T) / 2.
            qs.append(QuadraticFunction(P, q, r, relop))
        return cls(qs[0], qs[1:])
