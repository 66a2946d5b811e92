"""Drop-in facade: the reference's ``QCQP(prob).suggest()/improve()`` (qcqp/qcqp.py:367-432) on top of
the HIP engine.  Same method constants, same ``(f, v)`` return pairs, same sign convention for
maximisation, same kwargs and defaults; the inner loops run on the GPU for a whole POPULATION of
candidate points at once.

What is new (all optional, defaults reproduce the reference's one-point behaviour):
  suggest(..., num_samples=R, seed=s)   draw R candidates instead of one (resident on the GPU)
  improve(..., seed=s)                  seed of the counter-based stream used by phase 1 of
                                        coordinate descent (the reference uses the global NumPy RNG)
  qcqp.population_f / population_v      objective / max violation of every candidate
  qcqp.population()                     the candidates themselves, one per column

cvxpy is not needed (and not installed here): problems are given as ``qcqp_amd.Problem`` built from
raw ``(P, q, r, relop)`` arrays -- the data ``get_qcqp_form`` (utilities.py:318-347) would extract.
"""
import logging

import numpy as np

from . import settings as s
from .engine import E_UNSUPPORTED, Engine, EngineError
from .form import QCQPForm

# module logger (the reference configures the ROOT logger at import, qcqp.py:39, and logs per iteration; this engine
# never touches the root logger and reports per-run statistics instead: QCQP.last_stats and INFO records here)
log = logging.getLogger('qcqp_amd')

# improve(COORD_DESCENT): populations of at least this many points go through the lifecycle launch (see _improve)
STREAM_MIN = 512


class Variable(object):
    """Minimal stand-in for a cvxpy variable: ``size`` (rows, cols), ``value``, ``id``."""
    _next_id = 0

    def __init__(self, rows, cols=1):
        self.size = (int(rows), int(cols))
        self.value = None
        self.id = Variable._next_id
        Variable._next_id += 1


class _Objective(object):
    def __init__(self, name):
        self.NAME = name


class Problem(object):
    """A QCQP given by raw arrays.  ``funcs`` = [(P, q, r, relop), ...], objective first with
    relop None, written for the problem AS STATED (maximise problems give the function to be
    maximised; it is negated internally exactly like utilities.py:335-336)."""

    def __init__(self, funcs, maximize=False, var_sizes=None):
        funcs = list(funcs)
        n = int(np.asarray(funcs[0][1]).size)
        if maximize:
            P0, q0, r0, _ = funcs[0]
            funcs[0] = (-P0, -np.asarray(q0, dtype=np.float64), -float(r0), None)
        self.qcqp_form = QCQPForm.from_arrays(funcs)
        self.objective = _Objective('maximize' if maximize else 'minimize')
        sizes = var_sizes or [(n, 1)]
        assert sum(a * b for a, b in sizes) == n
        self._vars = [Variable(a, b) for a, b in sizes]

    @classmethod
    def from_minimize_form(cls, funcs, maximize=False, var_sizes=None):
        """``funcs`` already in minimise form (what QCQPForm holds), e.g. qcqp_amd.problems.*"""
        p = cls(funcs, False, var_sizes)
        p.objective = _Objective('maximize' if maximize else 'minimize')
        return p

    def variables(self):
        return self._vars


def assign_vars(xs, vals):
    """utilities.py:298-308"""
    if vals is None:
        for x in xs:
            x.value = np.full(x.size, np.nan)
    else:
        ind = 0
        for x in xs:
            size = x.size[0] * x.size[1]
            x.value = np.reshape(vals[ind:ind + size], x.size, order='F')
            ind += size


def flatten_vars(xs, n):
    """utilities.py:310-316 -- with the index advanced (the reference forgets ``ind += size`` and
    returns garbage past the first variable)."""
    ret = np.empty(n)
    ind = 0
    for x in xs:
        size = x.size[0] * x.size[1]
        ret[ind:ind + size] = np.ravel(x.value, order='F')
        ind += size
    return ret


class QCQP(object):
    def __init__(self, prob, device=0):
        if hasattr(prob, 'specs') and not hasattr(prob, 'qcqp_form'):
            # problems.GeneratedForm: the functions are synthesised on the device and exist nowhere else
            form = prob
            prob = Problem.__new__(Problem)
            prob.qcqp_form = form
            prob.objective = _Objective('minimize')
            prob._vars = [Variable(form.n, 1)]
        elif isinstance(prob, QCQPForm):
            form = prob
            prob = Problem.__new__(Problem)
            prob.qcqp_form = form
            prob.objective = _Objective('minimize')
            prob._vars = [Variable(form.n, 1)]
        if not hasattr(prob, 'qcqp_form'):
            # a cvxpy problem (cvxpy >= 1.0; the reference needs cvxpy 0.4's QuadCoeffExtractor): get_qcqp_form
            from .cvxpy_adapter import problem_from_cvxpy
            prob = problem_from_cvxpy(prob)
        self.prob = prob
        self.qcqp_form = prob.qcqp_form
        self.n = self.qcqp_form.n
        self.spectral_sol = None
        self.spectral_bound = None
        self.sdr_sol = None
        self.sdr_bound = None
        self.maximize_flag = (prob.objective.NAME == "maximize")
        self.engine = Engine(self.qcqp_form, device=device)
        self._resident = False      # the engine's population matches the variables' best point
        self.population_f = None
        self.population_v = None
        self.last_stats = None

    # ------------------------------------------------------------------ helpers
    def _sign(self, f):
        return -f if self.maximize_flag else f

    def _publish(self, f0, mv):
        """Record the population's values, write the best candidate into the variables and return
        the reference's (f, v) pair for it."""
        self.population_f = self._sign(np.asarray(f0))
        self.population_v = np.asarray(mv)
        idx, fb, vb, xb = self.engine.select_best(1e-4)
        assign_vars(self.prob.variables(), xb)
        self._assigned = np.array(xb, copy=True)
        self._resident = True
        self.best_index = idx
        return (self._sign(fb), vb)

    def population(self):
        return self.engine.download()

    def _resident_batches(self):
        """Is the engine's population still the K batches suggest(batches=K) drew (nobody wrote the variables since)?"""
        b = getattr(self, '_batches', None)
        return b is not None and self._resident and self.engine.pop_size == b[0] * b[1]

    # ------------------------------------------------------------------ suggest
    def suggest(self, method=s.RANDOM, eps=1e-8, *args, **kwargs):
        if method not in s.suggest_methods:
            raise Exception("Unknown suggest method: %s\n", method)
        R = int(kwargs.pop('num_samples', kwargs.pop('num_restarts', 1)))
        seed = kwargs.pop('seed', None)
        compat = kwargs.pop('compat', True)
        # batches = K: the reference's user loop `for ...: suggest(); improve()` (README.md:51-57) for K populations of
        # num_samples points at once -- batch b holds the global restart indices b R .. (b + 1) R - 1 of one keyed stream, i.e.
        # exactly the points K calls suggest(RANDOM, num_samples=R, first_index=b R) would draw; the improve() that follows
        # streams them through one persistent launch where the problem allows it (batch_results: the best point of each)
        keep_population = bool(kwargs.pop('keep_population', True))     # SDR, num_samples > 1: False = draw + evaluate without laying out the population
        K = int(kwargs.pop('batches', 1))
        first_index = int(kwargs.pop('first_index', 0))
        self._batches = None
        self.batch_results = None
        if K > 1:
            if method != s.RANDOM:
                raise Exception("suggest(batches=K) is defined for the RANDOM method")
            self._batches = (K, R, first_index)
            R = K * R
        if method == s.RANDOM:
            if R == 1 and seed is None and first_index == 0:
                x = np.random.randn(self.n)          # qcqp.py:382, same global-RNG draw
                self.engine.upload(x)
            else:
                self.engine.randn(R, seed=0 if seed is None else seed, first_index=first_index)
        elif method == s.SPECTRAL:
            if self.spectral_sol is None:
                # solve_spectral (qcqp.py:41-70): aggregated constraints, solved by the engine's own SDP solver
                from . import sdr as _sdr
                self.spectral_sol, bound, self.spectral_info = _sdr.solve_spectral(self.qcqp_form, seed=0 if seed is None else seed)
                self.spectral_bound = -bound if self.maximize_flag else bound      # qcqp.py:386-387
            self.engine.upload(np.asarray(self.spectral_sol, dtype=np.float64).ravel())
        elif method == s.SDR:
            if 'X' in kwargs:
                self.sdr_sol = np.asarray(kwargs.pop('X'), dtype=np.float64)
                self.sdr_bound = kwargs.pop('bound', None)
                if hasattr(self, 'mu'):
                    del self.mu
            if self.sdr_sol is None:
                # solve_sdr (qcqp.py:72-97): own solvers, heavy products on the device; every solve is certified
                # (dual slack PSD, feasibility, iteration limit) like the reference checks the solver status (qcqp.py:94-95)
                from . import sdr as _sdr
                sd = 0 if seed is None else seed
                sol = _sdr.solve_sdr(self.engine, self.qcqp_form, seed=sd)       # x_i^2 == d_i: mixing method, rigorous bound
                if sol is None and not self.engine.separable:
                    # any QCQP the dense path holds: Burer-Monteiro + augmented Lagrangian, matrices on the device
                    X, primal, info = _sdr.solve_sdr_general(self.engine, self.qcqp_form, seed=sd)
                    if self.n <= 4096:
                        lmin, S = _sdr.dual_certificate_device(self.engine, info['y'], info['yN'])
                        _sdr.certify(info, lmin, 1.0 + float(np.max(np.abs(S))), 'solve_sdr (general)')
                    else:
                        info['converged'] = None      # slack matrix too large to check on the host
                    # the dual value -y_N is the bound when the slack is PSD; otherwise only the primal value exists
                    sol = (X, info['dual_value'] if info.get('converged') else None, info)
                if sol is None:
                    fam = _sdr.separable_family(self.qcqp_form)
                    if fam is not None:
                        # boxes, discs, annuli, bounds on single coordinates: elementwise constraint operators
                        X, primal, info = _sdr.solve_sdr_separable(self.engine, self.qcqp_form, seed=sd)
                        lmin, S = _sdr.dual_slack_separable(self.qcqp_form, fam, info['y'], info['yN'])
                        _sdr.certify(info, lmin, 1.0 + float(np.max(np.abs(S))), 'solve_sdr (separable)')
                        sol = (X, info['dual_value'] if info['converged'] else None, info)
                if sol is None:
                    raise Exception("SDR suggest: no built-in SDP solver applies to this problem; pass "
                                    "suggest(SDR, X=...) or set qcqp.sdr_sol / qcqp.sdr_bound first.")
                self.sdr_sol, bound, self.sdr_info = sol
                # qcqp.py:392-393.  An uncertified solve publishes NO bound (the reference raises unless the solver reports
                # OPTIMAL, qcqp.py:94-95): the primal value <C, VV'> of an inexact factor is an upper estimate, kept only
                # in sdr_info['primal'].
                if bound is None:
                    self.sdr_bound = None
                    log.warning('solve_sdr: relaxation not certified (primal value %.8g kept in sdr_info); sdr_bound = None',
                                self.sdr_info.get('primal', float('nan')))
                else:
                    self.sdr_bound = -bound if self.maximize_flag else bound
                    log.info('solve_sdr: bound %.8g (primal %.8g, converged %s)', bound,
                             self.sdr_info.get('primal', bound), self.sdr_info.get('converged'))
            if not hasattr(self, 'mu'):
                X = np.asarray(self.sdr_sol, dtype=np.float64)
                self.mu = np.asarray(X[:-1, -1]).flatten()
                if compat:   # qcqp.py:395 as written: 1-D mu => element-wise square, row-broadcast
                    self.Sigma = X[:-1, :-1] - self.mu * self.mu.T + eps * np.eye(self.n)
                else:        # the intended X - mu mu^T + eps I
                    self.Sigma = X[:-1, :-1] - np.outer(self.mu, self.mu) + eps * np.eye(self.n)
                self._sdr_factor = None
            if R == 1 and seed is None:
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    x = np.random.multivariate_normal(self.mu, self.Sigma)   # qcqp.py:396
                self.engine.upload(x)
            else:
                if self._sdr_factor is None:
                    # the factor NumPy's multivariate_normal uses: x = mu + (xi * sqrt(s)) @ v
                    (u, sv, v) = np.linalg.svd(self.Sigma)
                    self._sdr_factor = np.ascontiguousarray((np.sqrt(sv)[:, None] * v).T)
                if not keep_population:
                    # draw + evaluate in one call, no population of R points (qcqpmi_sdr_sample_eval): the winner by the
                    # reference's `better` rule is re-drawn from its index and becomes the resident point, as after qcqp.py:398
                    sd = 0 if seed is None else seed
                    f0, mv = self.engine.sdr_sample_eval(self.mu, self._sdr_factor, R, seed=sd, first_index=first_index)
                    from .dist import select_best_host
                    best = int(select_best_host(f0, mv, 1e-4)[2])
                    self.engine.sdr_sample(None, None, 1, seed=sd, first_index=first_index + best)
                    fb, vb = self.engine.eval()
                    out = self._publish(fb, vb)
                    self.population_f, self.population_v, self.best_index = self._sign(np.asarray(f0)), np.asarray(mv), best
                    return out
                self.engine.sdr_sample(self.mu, self._sdr_factor, R, seed=0 if seed is None else seed, first_index=first_index)
        f0, mv = self.engine.eval()
        return self._publish(f0, mv)

    def _objective_factor(self, enable=True):
        """Once per problem: P0 = L L^T of low rank (a least-squares objective: rank = rows of A) -> the lifecycle kernel carries
        L^T X instead of multiplying with P0 (qcqp_amd.lowrank.objective_factor, qcqpmi_cd_set_objective_factor).  Tried for a
        dense-enough P0 with a positive diagonal of 256 <= n <= 4096; `factor=False` in improve() switches it off."""
        want = bool(enable)
        state = getattr(self, '_factor_state', None)
        if state is None:
            L = None
            f0 = self.qcqp_form.f0
            n = self.n
            if want and 256 <= n <= 4096:
                from .lowrank import objective_factor
                P0 = f0.P.toarray() if hasattr(f0.P, 'toarray') else np.asarray(f0.P)
                if np.all(np.diag(P0) > 0.0):
                    L = objective_factor(P0, max_rank=min(288, n // 2))
            self._factor_L = L
            state = self._factor_state = 'off'
        target = 'on' if (want and self._factor_L is not None) else 'off'
        if target != state:
            try:
                self.engine.cd_set_objective_factor(self._factor_L if target == 'on' else None)
                self._factor_state = target
            except EngineError as ex:
                if ex.code != E_UNSUPPORTED:
                    raise
                log.info('coord_descent: objective factor not used (%s)', ex)
                self._factor_L = None

    # ------------------------------------------------------------------ improve
    def _improve(self, method, *args, **kwargs):
        x0 = flatten_vars(self.prob.variables(), self.n)
        if not self._resident or not np.array_equal(x0, self._assigned):
            # the user (or another tool) wrote the variables: restart from that single point
            self.engine.upload(x0)
        if method == s.COORD_DESCENT:
            num_iters = kwargs.get('num_iters', 1000)
            viol_tol = kwargs.get('viol_tol', 1e-2)
            tol = kwargs.get('tol', 1e-4)
            phase1 = kwargs.get('phase1', True)
            seed = kwargs.get('seed', None)
            if seed is None:
                seed = int(np.random.randint(0, 2 ** 31 - 1))
            # reference_order=True: constraints that couple coordinates are walked in the reference's summation order
            # (slow; trajectories comparable with the reference value for value at any n -- qcqpmi_cd_reference_order)
            self.engine.cd_reference_order(bool(kwargs.get('reference_order', False)))
            first_index = int(kwargs.get('first_index', 0))
            batches = getattr(self, '_batches', None) if self._resident_batches() else None
            # stream = True / False forces / forbids the lifecycle launch (qcqpmi_cd_stream_run: phase 1, gate, phase 2 and the
            # evaluation of every restart inside ONE persistent kernel); default: from STREAM_MIN restarts on -- below that the
            # launch is as long as its slowest restart either way and the serial kernels are the simpler path.  Same restarts,
            # same points either way (tests/test_gpu_life.py); problems the kernel does not take fall back to qcqpmi_cd_run.
            stream = kwargs.get('stream', None)
            if stream is False:
                if batches is not None:       # the resident batches' global restart indices: same keyed draws as the streamed path
                    first_index = int(batches[2])
                batches = None
            elif batches is None and not kwargs.get('reference_order', False) and (stream or self.engine.pop_size >= STREAM_MIN):
                batches = (1, self.engine.pop_size, first_index)
            out = None
            if batches is not None:
                self._objective_factor(kwargs.get('factor', True))
                # population streaming: K batches of R restarts in ONE persistent launch (qcqpmi_cd_stream_run); families the
                # lifecycle kernel does not take run as one population of K R restarts -- the same restarts either way
                Kb, Rb, first_index = batches
                try:
                    out = self.engine.cd_stream_run(Kb, Rb, generate=False, phase1=phase1, num_iters=num_iters, viol_tol=viol_tol,
                                                    tol=tol, seed=seed, seed_stride=0, first_index=first_index, first_stride=Rb)
                except EngineError as ex:
                    if ex.code != E_UNSUPPORTED:     # (QCQPMI_EUNSUPPORTED: not the family of the lifecycle kernel)
                        raise
                    log.info('coord_descent: %s', ex)
            if out is None:
                out = self.engine.cd_run(phase1=phase1, num_iters=num_iters, viol_tol=viol_tol, tol=tol,
                                         seed=seed, first_index=first_index)
            if getattr(self, '_batches', None) is not None and self._resident_batches():
                from .dist import select_best_host
                Kb, Rb, _ = self._batches
                self.batch_results = []
                for b in range(Kb):
                    key = select_best_host(out['f0'][b * Rb:(b + 1) * Rb], out['maxviol'][b * Rb:(b + 1) * Rb], 1e-4)
                    i = key[2]
                    self.batch_results.append(dict(f=self._sign(out['f0'][b * Rb + i]), v=out['maxviol'][b * Rb + i], index=i))
            self.last_stats = dict(out, method=method, num_restarts=len(out['f0']),
                                   failed_restarts=int(np.count_nonzero(out['status1']) + np.count_nonzero(out['status2'] * (out['status1'] == 0))))
            log.info('coord_descent: %d restarts, phase-1 sweeps %.2f (max %d), phase-2 sweeps %.2f (max %.1f), accepted '
                     'updates %.1f, passed the violation gate %d, failed %d', len(out['f0']), out['sweeps1'].mean(),
                     int(out['sweeps1'].max()), out['visits2'].mean() / self.n, out['visits2'].max() / float(self.n),
                     out['accepted2'].mean(), int(out['ran_phase2'].sum()), self.last_stats['failed_restarts'])
            return self._publish(out['f0'], out['maxviol'])
        elif method == s.ADMM:
            return self._improve_admm(*args, **kwargs)
        elif method == s.DCCP:
            try:
                import dccp  # noqa: F401
            except ImportError:
                raise Exception("DCCP package is not installed.")
            raise Exception("improve(DCCP) delegates to an external solver and is out of scope of the HIP engine.")
        elif method == s.IPOPT:
            try:
                import pyipopt  # noqa: F401
            except ImportError:
                raise Exception("PyIpopt package is not installed.")
            raise Exception("improve(IPOPT) delegates to an external solver and is out of scope of the HIP engine.")

    def _improve_admm(self, *args, **kwargs):
        """improve_admm (qcqp.py:254-285): same kwargs, rho check / auto-rho; the iterations run on the GPU.
        Setup (what the reference gets from LAPACK / SuperLU):
          * constraint eigenpairs (utilities.py:160-162).  Default for constraints that couple coordinates: a
            rank-revealing range finder whose passes over the matrices run on the device (qcqp_amd.lowrank) -- if every
            constraint has rank <= 8 the iteration runs in the reduced bases (csrc/admm.h).  Otherwise, or with
            lowrank=False: full eigendecompositions, NumPy on the host like the reference (device_eigh=True:
            rocSOLVER's batched dsyevd on the device).
          * the z-update solve (qcqp.py:224-227): (2 (P0 + rho m I))^-1 is formed on the device (element-wise for a diagonal
            P0, otherwise a Newton-Schulz iteration on the engine's GEMM) and applied by the engine's GEMM;
          * lambda_min(P0) for the rho check / auto-rho (qcqp.py:262, 272): Lanczos with device products.
        host_setup=True computes both with NumPy/LAPACK where the reference does (bit-identical inputs for goldens)."""
        form = self.qcqp_form
        num_iters = kwargs.get('num_iters', 1000)
        viol_lim = kwargs.get('viol_lim', 1e4)
        tol = kwargs.get('tol', 1e-2)
        rho = kwargs.get('rho', None)
        phase1 = kwargs.get('phase1', True)
        host_setup = kwargs.get('host_setup', False)     # True: eigh / inv through NumPy exactly where the reference calls LAPACK
        P0 = np.asarray(form.f0.P.todense()) if hasattr(form.f0.P, 'todense') else np.asarray(form.f0.P)
        p0_diag = not np.any(P0 - np.diag(np.diag(P0)))
        if p0_diag:
            lmb_min = float(np.min(np.diag(P0)))
        elif host_setup:
            lmb_min = float(np.min(np.linalg.eigh(P0)[0]))   # qcqp.py:262, 272
        else:
            if getattr(form, '_lmb_min', None) is None:
                form._lmb_min = self.engine.p0_lambda_min()[0]     # Lanczos, products on the device
            lmb_min = form._lmb_min
        if rho is not None:
            if lmb_min + form.m * rho < 0:
                raise Exception("rho parameter is too small, need at least %.3f." % rho)
        else:
            if lmb_min < 0:
                rho = 2. * (1. - lmb_min) / form.m
            else:
                rho = 1. / form.m
            rho *= 50.
        if not getattr(self, '_eig_uploaded', False):
            mode = None
            if kwargs.get('device_eigh', False):
                # f.eigh for every constraint on the device (rocSOLVER batched dsyevd); needs the dense constraint
                # matrices resident.  The first use in a process loads the 0.9 GB librocsolver.so.
                self.engine.admm_setup()
                mode = 'rocsolver'
            elif self.engine.separable and kwargs.get('unit_bases', True) and form.unit_bases() is not None:
                # every constraint touches one coordinate: P_k = p e_i e_i^T, whose eigenvectors (utilities.py:160-162) are unit
                # vectors -- the basis is written down, no eigendecomposition (the reference: m calls of LAPACK on n x n
                # matrices), and the engine moves entries instead of multiplying by an n x m operator (qcqpmi_admm_unit_bases)
                lam, Bv, qhat = form.unit_bases()
                self.engine.admm_set_basis(lam, Bv, qhat)
                mode = 'unit bases (separable constraints)'
            elif kwargs.get('lowrank', True) and not self.engine.separable and form.n >= 64:
                from . import lowrank as _lr
                red = _lr.reduced_bases(self.engine, form, seed=0)
                if red is not None:
                    lam, Bv, qhat, info = red
                    self.engine.admm_set_basis(lam, Bv, qhat)
                    mode = 'reduced basis (rank <= %d, rp = %d)' % (int(info['rank'].max()), info['rp'])
            if mode is None:
                lm = np.zeros((form.m, form.n))
                Q = np.zeros((form.m, form.n, form.n))
                for k, f in enumerate(form.fs):
                    if f.eigh is None:   # cached like utilities.py:160-162
                        Pk = np.asarray(f.P.todense()) if hasattr(f.P, 'todense') else np.asarray(f.P)
                        f.eigh = np.linalg.eigh((Pk + Pk.T) / 2.)
                    lm[k], Q[k] = f.eigh
                self.engine.admm_set_eig(lm, Q)
                mode = 'full eigenbasis (host eigh)'
            self._eig_uploaded = True
            self._admm_mode = mode
            log.info('admm setup: %s', mode)
        if p0_diag:
            Minv = None
        elif host_setup:
            if form.rho != rho or form.z_solver is None:
                form.rho = rho
                form.z_solver = np.linalg.inv(2. * (P0 + rho * form.m * np.eye(form.n)))   # qcqp.py:224-227
            Minv = form.z_solver
            self._zsolver_rho = None      # the engine's device-side matrix is replaced by this one
        else:
            if getattr(self, '_zsolver_rho', None) != rho:
                res, its = self.engine.admm_zsolver_device(rho)     # Newton-Schulz on the engine's GEMM
                self._zsolver_rho = rho
                log.info('admm z-solver on the device: %d iterations, residual %.2e', its, res)
            Minv = None
        out = self.engine.admm_run(rho, Minv, phase1=phase1, num_iters=num_iters, tol=tol,
                                   viol_lim=viol_lim)
        self.last_stats = dict(out, method=s.ADMM, num_restarts=len(out['f0']), rho=rho, setup=getattr(self, '_admm_mode', None))
        log.info('admm: %d restarts, rho %.4g, phase-1 iterations %.1f (max %d), phase-2 iterations %.1f (max %d)',
                 len(out['f0']), rho, out['iters1'].mean(), int(out['iters1'].max()), out['iters2'].mean(),
                 int(out['iters2'].max()))
        return self._publish(out['f0'], out['maxviol'])

    def improve(self, method, *args, **kwargs):
        if not isinstance(method, list):
            methods = [method]
        else:
            methods = method
        if not all([method in s.improve_methods for method in methods]):
            raise Exception("Unknown improve method(s): ", methods)
        if any([x is None or x.value is None for x in self.prob.variables()]):
            self.suggest()
        for method in methods:
            try:
                f, v = self._improve(method, *args, **kwargs)
            except Exception:
                # the engine's population may have been advanced by the failed call: it no longer mirrors the
                # variables, the next call starts again from what the variables hold
                self._resident = False
                raise
        return (f, v)
