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
        self._unit_bases = False   # cache of unit_bases()
        self.z_solver = None  # for ADMM

    def unit_bases(self):
        """(lam (m, 1), Bv (m, 1, n), qhat (m, 1)) for constraints that each touch ONE coordinate: the nonzero eigenpair of
        P_k = p e_i e_i^T is (p, e_i), q_k = q e_i (utilities.py:160-166 in closed form).  None if some constraint couples coordinates."""
        if self._unit_bases is not False:
            return self._unit_bases
        form = self
        m, n = form.m, form.n
        lam = np.zeros((m, 1)); Bv = np.zeros((m, 1, n)); qhat = np.zeros((m, 1))
        for k, f in enumerate(form.fs):
            if sp.issparse(f.P):
                Pc = f.P.tocoo()
                keep = Pc.data != 0
                touched = set(Pc.row[keep].tolist()) | set(Pc.col[keep].tolist())
            else:
                rr, cc = np.nonzero(f.P)
                touched = set(rr.tolist()) | set(cc.tolist())
            touched |= set(np.nonzero(f.qarray)[0].tolist())
            if len(touched) > 1:
                self._unit_bases = None
                return None
            i = touched.pop() if touched else 0
            lam[k, 0] = f.P[i, i]
            Bv[k, 0, i] = 1.0
            qhat[k, 0] = f.qarray[i]
        self._unit_bases = (lam, Bv, qhat)
        return self._unit_bases

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
