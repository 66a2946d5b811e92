#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container: it imports /root/reference read-only (with the
four cvxpy modules the reference imports at module level stubbed out -- the hot
path never calls into them) from a temporary working directory (importing
qcqp.qcqp creates qcqp.log in the CWD), and stores INPUTS + EXPECTED OUTPUTS as
small .npz files.  Nothing of the reference's source travels; the fixtures are data.

Usage:  python tools/gen_golden.py            (writes tests/golden/*.npz)
"""
import os
import sys
import tempfile
import types
import warnings

import numpy as np
import scipy.sparse as sp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden')

if not os.path.isdir(REF):
    sys.exit('reference not present: golden fixtures can only be generated in the build container')

sys.path.insert(0, REPO)
from qcqp_amd import problems  # noqa: E402


def install_stubs():
    m = types.ModuleType('cvxpy')
    m.__path__ = []
    u = types.ModuleType('cvxpy.utilities')
    u.QuadCoeffExtractor = object
    lo = types.ModuleType('cvxpy.lin_ops')
    lo.__path__ = []
    lu = types.ModuleType('cvxpy.lin_ops.lin_utils')
    sys.modules.update({'cvxpy': m, 'cvxpy.utilities': u, 'cvxpy.lin_ops': lo,
                        'cvxpy.lin_ops.lin_utils': lu})
    sys.path.insert(0, REF)


os.chdir(tempfile.mkdtemp(prefix='qcqp_golden_'))
install_stubs()
import qcqp.utilities as U  # noqa: E402
import qcqp.qcqp as Q  # noqa: E402
import qcqp.settings as S  # noqa: E402

warnings.simplefilter('ignore')
RELCODE = {None: 0, '<=': 1, '==': 2}


def ref_prob(funcs):
    fs = []
    for (P, q, r, relop) in funcs:
        n = np.asarray(q).size
        fs.append(U.QuadraticFunction(sp.csr_matrix(P), sp.csc_matrix(np.asarray(q).reshape(n, 1)),
                                      r, relop))
    return U.QCQPForm(fs[0], fs[1:])


def pack_funcs(funcs):
    """Dense storage of a problem so that the fixture is self-contained."""
    n = np.asarray(funcs[0][1]).size
    P = np.stack([np.asarray(sp.csr_matrix(f[0]).todense()) for f in funcs])
    q = np.stack([np.asarray(f[1], dtype=float).ravel() for f in funcs])
    r = np.array([f[2] for f in funcs], dtype=float)
    rel = np.array([RELCODE[f[3]] for f in funcs], dtype=np.int64)
    return dict(P=P, q=q, r=r, relop=rel, n=np.int64(n))


def save(name, **kw):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **kw)
    print('wrote', name, {k: np.asarray(v).shape for k, v in kw.items()})


def small_problems():
    return {
        'bls10': problems.boolean_least_squares(10, 15, seed=1, legacy_seed=True),
        'bls32': problems.boolean_least_squares(32, 24, seed=3),
        'maxcut12': problems.maxcut(12, 0.5, seed=2),
        'dense16': problems.dense_indefinite(16, 5, seed=7),
        'beam10': problems.beamforming(5, 3, 2, seed=1),
    }


# ------------------------------------------------------------------ G1 / G2
def g1_g2():
    for name, (funcs, maxi, _) in small_problems().items():
        prob = ref_prob(funcs)
        n = prob.n
        rs = np.random.RandomState(11)
        X = rs.randn(n, 8)
        X[:, 1] = np.sign(X[:, 1])            # exactly feasible for x^2==1 families
        X[:, 2] = X[:, 1] * (1 + 3e-5)        # inside the first 1e-4 bucket
        X[:, 3] = X[:, 1] * (1 + 7e-5)        # second bucket for x^2==1
        F = np.array([[f.eval(X[:, s]) for s in range(8)] for f in [prob.f0] + prob.fs])
        V = np.array([[f.violation(X[:, s]) for s in range(8)] for f in prob.fs])
        maxviol = V.max(axis=0)
        # better(): all ordered pairs, incl. (a, a.copy()) ties
        B = np.zeros((8, 8), dtype=np.int64)
        for a in range(8):
            for b in range(8):
                xa, xb = X[:, a].copy(), X[:, b].copy()
                ret = prob.better(xa, xb)
                B[a, b] = 1 if ret is xa else 2
        # get_onevar_func
        ks = rs.randint(0, n, size=6)
        xs = rs.randn(6, n)
        T = np.zeros((6, prob.m + 1, 3))
        for t in range(6):
            for j, f in enumerate([prob.f0] + prob.fs):
                ov = f.get_onevar_func(xs[t], ks[t])
                T[t, j] = [ov.P, ov.q, ov.r]
        save('g1_' + name, X=X, F=F, V=V, maxviol=maxviol, better=B, ov_k=ks, ov_x=xs, ov_T=T,
             maximize=np.int64(maxi), **pack_funcs(funcs))


# ----------------------------------------------------------------------- G3
def g3():
    rows, outs = [], []
    vals_p = [2.0, 1.0, 1e-3, 1.5e-4, 1e-4, 5e-5, 0.0, -5e-5, -1e-4, -1.5e-4, -1.0, -3.0]
    vals_q = [0.0, 1.0, -2.5, 5e-5, -5e-5, 1e-4, 2e-4, -2e-4]
    vals_r = [-1.0, 0.0, 0.3, 2.0, -4.0]
    vals_s = [0.0, -1e-4, 5e-5, 1e-2, 0.7]
    for relop in ('<=', '=='):
        for p in vals_p:
            for q in vals_q:
                for r in vals_r:
                    for s in vals_s:
                        f = U.OneVarQuadraticFunction(p, q, r, relop)
                        I = U.get_feasible_intervals(f, s)
                        o = np.full(9, np.nan)
                        o[0] = len(I)
                        for i, iv in enumerate(I):
                            o[1 + 2 * i], o[2 + 2 * i] = iv
                        rows.append([p, q, r, s, RELCODE[relop]])
                        outs.append(o)
    save('g3_intervals', cases=np.array(rows), out=np.array(outs))


# ----------------------------------------------------------------------- G4
def g4():
    cases = []

    def run(f0, fs, s, seed):
        np.random.seed(seed)
        pos0 = np.random.get_state()[2]
        obj = U.OneVarQuadraticFunction(*f0)
        nfs = [U.OneVarQuadraticFunction(*f) for f in fs]
        try:
            x = U.onevar_qcqp(obj, nfs, s)
            err = 0
        except Exception as e:  # OverflowError / NameError quirks
            x, err = None, 1
        st = np.random.get_state()
        draws = (st[2] - pos0) % 624 if err == 0 else -1
        cases.append(dict(f0=f0, fs=fs, s=s, seed=seed, x=x, err=err, draws=draws))

    # hand-picked: unique minimiser inside, endpoint, None, zero-width (A.5), coincident endpoints (A.6)
    run((1., -3., 0.), [(1., 0., -1., '==')], 0.0, 1)
    run((1., -3., 0.), [(1., 0., -1., '==')], 1e-4, 1)
    run((1., 0.2, 0.), [(1., 0., -4., '<=')], 0.0, 2)
    run((1., 9.0, 0.), [(1., 0., -4., '<=')], 0.0, 2)
    run((1., 0., 0.), [(0., 1., -1., '<='), (1., 0., -1., '<=')], 0.0, 3)
    run((1., 0., 0.), [(0., 1., -1.5, '<='), (1., 0., -1., '<=')], 0.0, 3)
    run((-1., 0.5, 2.), [(1., 0., -1., '<=')], 0.5, 4)
    run((-1., 0.0, 2.), [(1., 0., -1., '<=')], 0.0, 4)        # exact tie between endpoints -> choice
    run((0., 1., 0.), [(-1., 0., 1., '<='), (1., 0., -9., '<=')], 0.0, 5)
    run((0., 0., 0.), [(1., 0., -1., '==')], 1e-3, 6)          # objective zero -> uniform in random interval
    run((0., 0., 0.), [(1., 0., -1., '==')], -1e-5, 6)         # infeasible -> None
    run((0., 0., 0.), [(1., 0., -1., '<=')], 0.2, 7)
    run((0., 0., 0.), [(0., 1., 0., '<=')], 0.0, 7)            # unbounded interval -> OverflowError
    run((0., -1., 0.), [(0., 1., 0., '<=')], 0.0, 8)           # unbounded ray, linear objective
    run((1., 0., 0.), [(5e-5, 5e-5, 10., '<=')], 0.0, 9)       # A.7 "always feasible"
    run((1., 2., 0.), [], 0.3, 9)                               # no constraints -> whole line
    rs = np.random.RandomState(77)
    for t in range(160):
        k = rs.randint(1, 5)
        fs = []
        for j in range(k):
            relop = '<=' if rs.rand() < 0.6 else '=='
            p = rs.choice([rs.randn(), 0.0, 1.0, -1.0], p=[0.6, 0.1, 0.15, 0.15])
            fs.append((float(p), float(rs.randn()), float(rs.randn() - 1.0), relop))
        zero_obj = rs.rand() < 0.3
        f0 = (0., 0., 0.) if zero_obj else (float(rs.randn()), float(rs.randn()), float(rs.randn()))
        run(f0, fs, float(abs(rs.randn()) * rs.choice([0.01, 0.5, 2.0])), 100 + t)
    K = max(len(c['fs']) for c in cases)
    N = len(cases)
    f0 = np.array([c['f0'] for c in cases])
    fs = np.full((N, K, 4), np.nan)
    nf = np.zeros(N, dtype=np.int64)
    for i, c in enumerate(cases):
        nf[i] = len(c['fs'])
        for j, f in enumerate(c['fs']):
            fs[i, j] = [f[0], f[1], f[2], RELCODE[f[3]]]
    save('g4_onevar_qcqp', f0=f0, fs=fs, nf=nf, s=np.array([c['s'] for c in cases]),
         seed=np.array([c['seed'] for c in cases]),
         x=np.array([np.nan if c['x'] is None else c['x'] for c in cases]),
         isnone=np.array([c['x'] is None for c in cases]),
         err=np.array([c['err'] for c in cases]), draws=np.array([c['draws'] for c in cases]))


# ----------------------------------------------------------------------- G5
def g5():
    rs = np.random.RandomState(5)
    n = 12
    mats = {}
    G = rs.randn(n, n)
    mats['indef'] = (G + G.T) / 2
    mats['psd'] = G.dot(G.T) / n
    mats['nsd'] = -G.dot(G.T) / n
    a, b = rs.randn(n), rs.randn(n)
    mats['rank2p'] = np.outer(a, a) + np.outer(b, b)
    mats['rank2n'] = -(np.outer(a, a) + np.outer(b, b))
    e = np.zeros((n, n)); e[3, 3] = 1.0
    mats['eiei'] = e
    recs = []
    for name, P in mats.items():
        for relop in ('<=', '=='):
            for trial in range(3):
                q = rs.randn(n) * (0.0 if name == 'eiei' else 1.0)
                r = {'indef': -0.5, 'psd': -2.0, 'nsd': 3.0, 'rank2p': -2.0, 'rank2n': 20.0,
                     'eiei': -1.0}[name]
                z = rs.randn(n) * (1 + trial)
                f = U.QuadraticFunction(sp.csr_matrix(P), sp.csc_matrix(q.reshape(n, 1)), r, relop)
                x = U.onecons_qcqp(z.copy(), f)
                early = (relop == '<=' and f.eval(z) <= 0)
                if f.eigh is None:
                    f.eigh = np.linalg.eigh(np.asarray(((f.P + f.P.T) / 2.).todense()))
                recs.append(dict(P=P, q=q, r=r, relop=RELCODE[relop], z=z, x=np.asarray(x).ravel(),
                                 lmb=np.asarray(f.eigh[0]), Q=np.asarray(f.eigh[1]), early=early,
                                 fx=f.eval(np.asarray(x).ravel())))
    save('g5_onecons', **{k: np.array([r_[k] for r_ in recs]) for k in recs[0]})


# ------------------------------------------------------------------ G6 / G7
def g6_g7():
    fams = {
        'bls10': problems.boolean_least_squares(10, 15, seed=1, legacy_seed=True),
        'bls32': problems.boolean_least_squares(32, 24, seed=3),
        'bls64': problems.boolean_least_squares(64, 48, seed=4),
        'maxcut12': problems.maxcut(12, 0.5, seed=2),
        'dense16': problems.dense_indefinite(16, 5, seed=7),
        'dense32': problems.dense_indefinite(32, 8, seed=7),
        # examples/circle_packing.py: two variables (centres 2 x 5, radius), sparse separation constraints; minimise form
        'circle5': (problems.circle_packing(5, minimize_form=True)[0], True, {}),
    }
    only = [a_[7:] for a_ in sys.argv[1:] if a_.startswith('--only=')]
    for name, (funcs, maxi, _) in fams.items():
        if only and name not in only:
            continue
        prob = ref_prob(funcs)
        n = prob.n
        rs = np.random.RandomState(21)
        R = 4
        # phase 2 from feasible-within-slack starts: x0 = sign*(1+delta)
        if name.startswith('bls') or name.startswith('maxcut'):
            X0 = np.sign(rs.randn(n, R)) * (1 + 2e-5 * rs.rand(n, R))
        elif name.startswith('circle'):
            X0 = np.vstack([1.0 + 8.0 * rs.rand(n - 1, R), 0.05 + 0.1 * rs.rand(1, R)])     # centres in the box, a small radius
        else:
            X0 = 0.05 * rs.randn(n, R)
        p2_x, p2_seed = [], []
        for r in range(R):
            np.random.seed(500 + r)
            p2_x.append(Q.coord_descent_phase2(X0[:, r], prob))
            p2_seed.append(500 + r)
        # phase 1 and the full driver from randn starts, seeded global RNG
        Y0 = rs.randn(n, R)
        p1_x, full_x, p1_pos, full_pos = [], [], [], []
        for r in range(R):
            np.random.seed(900 + r)
            p1_x.append(Q.coord_descent_phase1(Y0[:, r], prob, num_iters=30))
            p1_pos.append(np.random.get_state()[2])
            np.random.seed(900 + r)
            full_x.append(Q.improve_coord_descent(Y0[:, r], prob, num_iters=30))
            full_pos.append(np.random.get_state()[2])
        p2_x, p1_x, full_x = map(np.array, (p2_x, p1_x, full_x))
        ev = lambda xs: np.array([[prob.f0.eval(x), max(prob.violations(x))] for x in xs])
        save('g6_cd_' + name, X0=X0, p2_x=p2_x.T, p2_fv=ev(p2_x), p2_seed=np.array(p2_seed),
             Y0=Y0, p1_x=p1_x.T, p1_fv=ev(p1_x), p1_pos=np.array(p1_pos),
             full_x=full_x.T, full_fv=ev(full_x), full_pos=np.array(full_pos),
             seed0=np.int64(900), num_iters=np.int64(30), maximize=np.int64(maxi),
             **pack_funcs(funcs))


# ----------------------------------------------------------------------- G8
def g8():
    fams = {
        'beam10': (problems.beamforming(5, 3, 2, seed=1), np.sqrt(5.)),
        'beam40': (problems.beamforming(20, 5, 2, seed=1), np.sqrt(7.)),
        'bls10': (problems.boolean_least_squares(10, 15, seed=1, legacy_seed=True), None),
        'dense16': (problems.dense_indefinite(16, 5, seed=7), 4.0),
    }
    for name, ((funcs, maxi, _), rho) in fams.items():
        prob = ref_prob(funcs)
        n = prob.n
        rs = np.random.RandomState(31)
        x0 = rs.randn(n)
        iters = 60
        z1 = Q.admm_phase1(x0, prob, 1e-2, iters)
        rho_used = rho
        if rho_used is None:
            lm = np.linalg.eigh(prob.f0.P.todense())[0]
            lmin = lm.min()
            rho_used = 50. * (2. * (1. - lmin) / prob.m if lmin < 0 else 1. / prob.m)
        z2 = Q.admm_phase2(z1, prob, rho_used, 1e-2, iters, 1e4)
        prob2 = ref_prob(funcs)
        xa = Q.improve_admm(x0, prob2, num_iters=iters, rho=rho)
        eigs = [f.eigh if f.eigh is not None else np.linalg.eigh(np.asarray(((f.P + f.P.T) / 2.).todense()))
                for f in prob.fs]
        save('g8_admm_' + name, x0=x0, z1=z1, z2=z2, xa=np.asarray(xa).ravel(), rho=np.float64(rho_used),
             rho_arg=np.float64(np.nan if rho is None else rho), iters=np.int64(iters),
             fv1=np.array([prob.f0.eval(z1), max(prob.violations(z1))]),
             fv2=np.array([prob.f0.eval(z2), max(prob.violations(z2))]),
             fva=np.array([prob.f0.eval(xa), max(prob.violations(xa))]),
             lmb=np.array([np.asarray(e[0]) for e in eigs]), Q=np.array([np.asarray(e[1]) for e in eigs]),
             **pack_funcs(funcs))


# ------------------------------------------------------------------ G9 / G10
class FakeVar(object):
    def __init__(self, n, vid):
        self.size = (n, 1)
        self.value = None
        self.id = vid


class FakeObjective(object):
    def __init__(self, name):
        self.NAME = name


class FakeProb(object):
    def __init__(self, n, maximize):
        self._v = [FakeVar(n, 0)]
        self.objective = FakeObjective('maximize' if maximize else 'minimize')

    def variables(self):
        return self._v


def make_handler(funcs, maximize):
    h = object.__new__(Q.QCQP)
    h.qcqp_form = ref_prob(funcs)
    h.n = h.qcqp_form.n
    h.prob = FakeProb(h.n, maximize)
    h.spectral_sol = h.spectral_bound = None
    h.sdr_sol = h.sdr_bound = None
    h.maximize_flag = maximize
    return h


def lifted_solution(n, rank, mu_scale, rs):
    """A PSD (n+1)x(n+1) matrix with X[-1,-1] = 1 standing in for the SDP optimum."""
    V = rs.randn(n + 1, rank)
    V[-1] = 0
    V[-1, 0] = 1.0
    V[:-1, 0] = mu_scale * rs.randn(n)
    X = V.dot(V.T)
    return X


def g9_g10():
    rs = np.random.RandomState(41)
    for name, maxi in (('bls10', False), ('maxcut12', True)):
        funcs = small_problems()[name][0]
        n = np.asarray(funcs[0][1]).size
        X = lifted_solution(n, 4, 0.0 if maxi else 0.4, rs)
        h = make_handler(funcs, maxi)
        h.sdr_sol = np.asmatrix(X)       # cvxpy 0.4 returns np.matrix
        h.sdr_bound = -1.0
        # first suggest(SDR) call with sdr_sol preset skips solve_sdr AND the mu/Sigma lines, so set them
        # the way qcqp.py:394-395 does
        h.mu = np.asarray(h.sdr_sol[:-1, -1]).flatten()
        h.Sigma = h.sdr_sol[:-1, :-1] - h.mu * h.mu.T + 1e-8 * sp.identity(n)
        xs, fvs = [], []
        np.random.seed(7)
        for t in range(5):
            f, v = h.suggest(S.SDR)
            xs.append(np.asarray(h.prob.variables()[0].value).ravel(order='F'))
            fvs.append([f, v])
        save('g9_sdr_' + name, X=X, mu=h.mu, Sigma=np.asarray(h.Sigma), xs=np.array(xs).T,
             fv=np.array(fvs), seed=np.int64(7), maximize=np.int64(maxi), **pack_funcs(funcs))

    # G10: API-level flow on the README data
    funcs, maxi, _ = problems.boolean_least_squares(10, 15, seed=1, legacy_seed=True)
    h = make_handler(funcs, False)
    np.random.seed(42)
    out = []
    fv = h.suggest(S.RANDOM)
    out.append(list(fv)); x_rand = np.asarray(h.prob.variables()[0].value).ravel(order='F')
    fv = h.improve(S.COORD_DESCENT)
    out.append(list(fv)); x_cd = np.asarray(h.prob.variables()[0].value).ravel(order='F')
    fv = h.improve([S.COORD_DESCENT, S.ADMM], phase1=False, num_iters=50)
    out.append(list(fv)); x_chain = np.asarray(h.prob.variables()[0].value).ravel(order='F')
    pos = np.random.get_state()[2]
    save('g10_api_bls10', fv=np.array(out), x_rand=x_rand, x_cd=x_cd, x_chain=x_chain,
         seed=np.int64(42), pos=np.int64(pos), **pack_funcs(funcs))


if __name__ == '__main__':
    if any(a_.startswith('--only=') for a_ in sys.argv[1:]):      # --only=circle5: one family of G6 / G7
        g6_g7()
        sys.exit(0)
    g1_g2()
    g3()
    g4()
    g5()
    g6_g7()
    g8()
    g9_g10()
