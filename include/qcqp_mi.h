/*
 * qcqp_mi.h -- C ABI of the MI355X-native Suggest-and-Improve engine (libqcqp_mi.so).
 *
 * This is the drop-in boundary for the hot path of cvxgrp/qcqp.  The reference has no FFI
 * or plugin registry; its seam is the plain-Python call sites
 *     QCQP.suggest            qcqp/qcqp.py:378-401   (RANDOM :381-382, SDR tail :394-401)
 *     QCQP._improve           qcqp/qcqp.py:403-417   (improve_coord_descent :181-192,
 *                                                      improve_admm :254-285)
 * operating on a QCQPForm (qcqp/utilities.py:122-146) of QuadraticFunction objects
 * (qcqp/utilities.py:41-62).  Each entry point below names the reference code it replaces.
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative QCQPMI_E* code and
 *     never throws; qcqpmi_last_error() returns the message for the last failure.
 *   - the caller owns every host buffer (C-contiguous, fp64 / int64) and may free it on return.
 *   - the context owns all device memory and one HIP stream; a context is not thread-safe,
 *     distinct contexts are independent.
 *   - candidate points are handed over as an n x R "column per candidate" array:
 *     candidate r occupies X[r*n .. r*n + n).
 *   - "population" = the device-resident batch of candidates (restarts of improve(), samples of
 *     suggest()); the timed hot path works on the resident population, host copies are explicit.
 *   - relop codes: 0 = none (objective), 1 = '<=', 2 = '=='.
 *   - all arithmetic is fp64.
 */
#ifndef QCQP_MI_H
#define QCQP_MI_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QCQPMI_ABI_VERSION 6

enum {
    QCQPMI_OK = 0,
    QCQPMI_EINVAL = -1,       /* bad argument */
    QCQPMI_EHIP = -2,         /* HIP runtime failure (no device, OOM, launch error) */
    QCQPMI_ESTATE = -3,       /* call order (e.g. run before finalize / without population) */
    QCQPMI_EUNSUPPORTED = -4, /* problem structure not handled by the HIP engine yet */
    QCQPMI_EREFERENCE = -5,   /* the reference would raise on this input (see last_error) */
    QCQPMI_ECOMM = -6         /* RCCL failure */
};

enum { QCQPMI_FMT_DENSE = 0, QCQPMI_FMT_CSR = 1 };

typedef struct qcqpmi_ctx qcqpmi_ctx;

/* ---- library / device ------------------------------------------------------------------ */
int qcqpmi_abi_version(void);
int qcqpmi_device_count(void); /* number of visible HIP devices, 0 if none, never fails */
const char *qcqpmi_last_error(const qcqpmi_ctx *ctx); /* ctx may be NULL: last create error */

/* ---- context = one QCQPForm on one GPU  (utilities.py:122-130) ------------------------- */
int qcqpmi_ctx_create(qcqpmi_ctx **out, int64_t n, int64_t m, int device);
void qcqpmi_ctx_destroy(qcqpmi_ctx *ctx);

/* QuadraticFunction(P, q, r, relop)  (utilities.py:41-46).  k = 0 is the objective f0,
 * k = 1..m the constraints fs[k-1].  P must be symmetric (get_qcqp_form symmetrises,
 * utilities.py:333,345).  DENSE: vals = n*n row-major, idx = ptr = NULL, nnz ignored.
 * CSR: ptr[n+1], idx[nnz], vals[nnz].
 * Constraints that couple coordinates are kept row-major on the device as long as all of them fit 16 GB (the paths in the
 * reference's arithmetic and the device-side ADMM setup read them there); beyond that -- BASELINE.json configs[4] is 137 GB --
 * every coupled function is packed for the matrix cores when this call receives it (streaming upload: no second copy on
 * the host or the device; up to 200 GB of packed matrices) and the problem runs on the dense path only. */
int qcqpmi_set_quad(qcqpmi_ctx *ctx, int64_t k, int format, const double *vals,
                    const int64_t *idx, const int64_t *ptr, int64_t nnz, const double *q,
                    double r, int relop);
/* Synthetic dense function generated ON THE DEVICE, in place of qcqpmi_set_quad for function k
 * (BASELINE.json configs[4]: 1025 dense 4096 x 4096 matrices = 137.6 GB never exist on the host;
 * SURVEY.md section 8(d) cfg5 "generate on device with Philox from the seed"):
 *     P_k[i][j] = scale * w_ij * N(seed, k, min(i,j) n + max(i,j)) + diag_add * [i == j],
 *     w_ii = 1, w_ij = 1/sqrt(2)            (the law of (G + G^T)/2 with G_ij ~ N(0,1), symmetric)
 *     q_k[j]    = qscale * N'(seed, k, j)
 * N, N' = the engine's keyed Philox normal (the generator of qcqpmi_pop_randn; oracle:
 * orc_keyed_normal(seed, 2^48 + k, elem) and orc_keyed_normal(seed, 2^49 + k, j)). */
int qcqpmi_set_quad_generated(qcqpmi_ctx *ctx, int64_t k, uint64_t seed, double scale, double qscale,
                              double diag_add, double r, int relop);
/* Build the device layouts (MFMA-packed P0, per-coordinate constraint lists). */
int qcqpmi_finalize(qcqpmi_ctx *ctx);
/* 1 if every constraint touches exactly one coordinate (Boolean / box families) */
int qcqpmi_is_separable(const qcqpmi_ctx *ctx);

/* ---- population ------------------------------------------------------------------------- */
int qcqpmi_pop_upload(qcqpmi_ctx *ctx, const double *X, int64_t R);
int qcqpmi_pop_download(qcqpmi_ctx *ctx, double *X, int64_t R);
int64_t qcqpmi_pop_size(const qcqpmi_ctx *ctx);
/* suggest(RANDOM) batched: x_r = randn(n)  (qcqp.py:381-382).  Counter-based stream keyed on
 * (seed, first_index + r, element): independent of the sharding over GPUs. */
int qcqpmi_pop_randn(qcqpmi_ctx *ctx, int64_t R, uint64_t seed, uint64_t first_index);
/* suggest(SDR) tail batched: x_s = mu + F xi_s, xi_s ~ N(0, I)  (qcqp.py:396), where
 * F (n x n row-major) is any factor with F F^T = Sigma (the host computes it once: Cholesky, or
 * the SVD factor NumPy's multivariate_normal uses).  Xi (n x S, column per sample) may be given
 * for reproducibility tests; NULL = device Philox normals keyed on (seed, first_index + s).
 * mu == NULL and F == NULL: draw again from the pair of the previous call, which is still resident and packed (no
 * upload; many batches from one relaxation).  Returns without synchronising: the samples are in stream order. */
int qcqpmi_pop_sdr_sample(qcqpmi_ctx *ctx, const double *mu, const double *F, int64_t S,
                          uint64_t seed, uint64_t first_index, const double *Xi);

/* ---- evaluation: QuadraticFunction.eval / violation, QCQPForm.violations
 *      (utilities.py:49-62, 133-134; callers qcqp.py:399-401, 415-417) -------------------- */
/* on the resident population; f0/maxviol have R entries; F (optional) is (m+1) x R row-major */
int qcqpmi_pop_eval(qcqpmi_ctx *ctx, double *f0, double *maxviol, double *F);
/* suggest(SDR) for S samples in ONE call -- draw and evaluate (qcqp.py:396, 399, 401; SURVEY.md section 8b's
 * qcqpmi_sdr_sample_eval): x_s = mu + F xi_s with device Philox normals keyed on (seed, first_index + s), f0[s] = f0(x_s),
 * maxviol[s] = max_k violation_k(x_s).  Chunks of samples are drawn and evaluated back to back in two buffers that every chunk
 * reuses (<= 32 MB each): no population of S points is laid out in HBM.  X_opt (n x S, column per sample, or NULL): the points;
 * with NULL they are not kept -- re-draw the winner from its index (S = 1, first_index + s).  mu == F == NULL: the pair of the
 * previous call.  Afterwards the resident population is the last chunk.  Synchronises before it returns. */
int qcqpmi_sdr_sample_eval(qcqpmi_ctx *ctx, const double *mu, const double *F, int64_t S, uint64_t seed,
                           uint64_t first_index, double *X_opt, double *f0, double *maxviol);
/* host convenience: upload X, evaluate */
int qcqpmi_eval_batch(qcqpmi_ctx *ctx, const double *X, int64_t S, double *f0, double *maxviol,
                      double *F);

/* ---- improve(COORD_DESCENT) on the resident population
 *      (improve_coord_descent qcqp.py:181-192; phase 1 :101-148; phase 2 :152-178;
 *       onevar_qcqp utilities.py:241-288; get_feasible_intervals utilities.py:198-232).
 * Outputs (each R entries, any may be NULL):
 *   sweeps1/sweeps2   sweeps started in phase 1 / phase 2
 *   visits2           coordinate visits of phase 2 (loop bodies of qcqp.py:162)
 *   accepted2         accepted coordinate updates of phase 2
 *   ran_phase2        1 if the restart passed the max-violation gate (qcqp.py:189)
 *   f0/maxviol        objective / max violation of the final point (qcqp.py:415-417)       */
int qcqpmi_cd_run(qcqpmi_ctx *ctx, int phase1, int64_t num_iters, double viol_tol, double tol,
                  uint64_t seed, uint64_t first_index, int64_t *sweeps1, int64_t *sweeps2,
                  int64_t *visits2, int64_t *accepted2, uint8_t *ran_phase2, double *f0,
                  double *maxviol);
/* The same run in stages, for callers that keep two contexts (two HIP streams) on one GPU and overlap the preparation of
 * the next population -- suggest, phase 1, evaluation, gate -- with the phase-2 kernel of the current one, whose tail leaves
 * most CUs idle (bench.py).  stage 0 = qcqpmi_cd_run; 1 = phase 1 + evaluation + gate, asynchronous; 2 = launch of phase 2,
 * asynchronous; 3 = results, blocking.  Stages 1..3 must be called in order with the same parameters; the output pointers
 * are only used by stage 3 (and 0). */
int qcqpmi_cd_run_stage(qcqpmi_ctx *ctx, int stage, int phase1, int64_t num_iters, double viol_tol, double tol,
                        uint64_t seed, uint64_t first_index, int64_t *sweeps1, int64_t *sweeps2,
                        int64_t *visits2, int64_t *accepted2, uint8_t *ran_phase2, double *f0,
                        double *maxviol);
/* Per-restart status codes of the last qcqpmi_cd_run (R ints each; 0 = fine).  A restart on which the
 * reference would raise -- -1: np.random.uniform on an unbounded interval (utilities.py:267), -2: NameError in
 * OneVarQuadraticFunction.eval (utilities.py:119), -3: max() of an empty list (qcqp.py:117), -4: feasible set
 * with more segments than the kernel holds -- is reported here, its f0 / maxviol are +inf (it never wins
 * qcqpmi_select_best) and the other restarts are unaffected.  qcqpmi_cd_run itself returns
 * QCQPMI_EREFERENCE / QCQPMI_EUNSUPPORTED only when EVERY restart failed (R = 1: the reference's behaviour). */
int qcqpmi_cd_status(qcqpmi_ctx *ctx, int *status1, int *status2);

/* ---- unit operators: the reference's module-level helpers for a batch of independent cases, evaluated on the
 * device by the same code the coordinate-descent kernels inline (no context needed).
 *   get_feasible_intervals(f, s)  utilities.py:198-232:  pqrs = count x (p, q, r, s), relop per case;
 *        out = count x (n, lo0, hi0, lo1, hi1), at most two intervals (+-inf as IEEE infinities).
 *   onevar_qcqp(f0, fs, s)        utilities.py:241-288:  f0 = count x (p, q, r); fs = count x 4 x (p, q, r, relop as a
 *        number); nf[i] <= 4 constraints in use; status 1 = a point in x[i], 0 = None, < 0 = the reference raises
 *        (-1 OverflowError in np.random.uniform, -2 NameError in OneVarQuadraticFunction.eval); ties and the
 *        zero-objective draw (utilities.py:266-267, 288) use the keyed Philox stream (seed; case index).
 *        C_out (optional) = count x 5 x (lo, hi): the feasible set after the end-point sweep, nC_out its size. */
int qcqpmi_feasible_intervals_batch(int device, int64_t count, const double *pqrs, const int *relop, double *out);
/*   QuadraticFunction.get_onevar_func(x, k)  utilities.py:99-105: for every function j = 0..m of the problem (objective
 *        first) and every restart r of the resident population, the coefficients of f_j as a function of the coordinate
 *        coord[r] alone: out[(r (m + 1) + j) 3 + (0, 1, 2)] = (t2, t1, t0) = (P[k,k], 2 P[k,:] z + q[k], f_j(z)) with z = x_r,
 *        z[k] = 0, summed in the reference's order by the device functions the coupled-constraint kernel inlines
 *        (csrc/cd_general.h).  Needs the row-major matrices (not device-generated / streamed problems). */
int qcqpmi_onevar_coeffs(qcqpmi_ctx *ctx, const int64_t *coord, double *out);
int qcqpmi_onevar_qcqp_batch(int device, int64_t count, const double *f0, const double *fs, const int *nf,
                             const double *s, uint64_t seed, double *x, int *status, double *C_out, int *nC_out);

/* ---- improve(ADMM) on the resident population (improve_admm qcqp.py:254-285; admm_phase1 :195-212;
 * admm_phase2 :215-251; onecons_qcqp utilities.py:149-196).  The host supplies what the reference
 * obtains from LAPACK / SuperLU: the eigendecomposition of every constraint matrix (lmb: m x n,
 * Q: m x n x n, NumPy eigh layout: Q[k][:, j] = eigenvector j; cached on the context like f.eigh,
 * utilities.py:160-162) and, for phase 2, Minv = (2 (P0 + rho m I))^-1 (n x n) in place of the
 * SuperLU factorisation of qcqp.py:224-227.  rho must already be validated / chosen by the caller
 * (qcqp.py:261-278).  Minv may be NULL when P0 is diagonal (the inverse is formed on the device).  Outputs (R entries
 * each, may be NULL): iterations of phase 1 / phase 2, objective and max violation of the returned points. */
int qcqpmi_admm_set_eig(qcqpmi_ctx *ctx, const double *lmb, const double *Q);
/* The same cache computed ON THE DEVICE (f.eigh = LA.eigh(P), utilities.py:160-162, for every constraint at once:
 * rocSOLVER batched dsyevd on the resident dense matrices) -- for problems whose constraints couple coordinates.
 * Eigenvectors of degenerate eigenvalues span the same spaces as LAPACK's but are a different basis: iterates
 * agree with qcqpmi_admm_set_eig to rounding, not bit for bit. */
int qcqpmi_admm_setup(qcqpmi_ctx *ctx);
/* REDUCED bases for low-rank constraints (beamforming: rank 2).  onecons_qcqp (utilities.py:149-196) moves a point only
 * inside span(B_k), B_k = [eigenvectors of the nonzero eigenvalues of P_k, the unit vector along the part of q_k outside
 * them (eigenvalue 0)]: with lam (m x rp), Bv (m x rp x n, basis vector j of constraint k contiguous, orthonormal, zero
 * padded) and qhat = B_k^T q_k (m x rp) the ADMM iteration runs on (m rp) x n operators instead of (m n) x n --
 * algebraically the same iteration (duals live in span(B_k)).  The multiplier bracket comes from the nonzero
 * eigenvalues given here, not from LAPACK's round-off eigenvalues of the null space: results agree
 * with the full-basis path to the accuracy of the reference's own bisection (1e-6 on every multiplier). */
int qcqpmi_admm_set_basis(qcqpmi_ctx *ctx, int64_t rp, const double *lam, const double *Bv, const double *qhat);
/* Optional: the multiplier bracket [slo_k, ehi_k] (m values each, host) that the bisection of onecons_qcqp starts from
 * (utilities.py:176-186), replacing the one qcqpmi_admm_set_eig / _set_basis / _setup derived from their own eigenvalues;
 * -inf / +inf = "no eigenvalue of that sign": the reference's doubling search from -1 / +1.  The reference brackets with
 * EVERY eigenvalue LAPACK returns, round-off eigenvalues of the null space included (a rank-2 PSD matrix has n - 2
 * eigenvalues ~ +-1e-15, so ehi = -1 / min(lmb) ~ 1e14 instead of the doubling search); with the reference's
 * bracket the reduced-basis path visits the reference's midpoints and returns its multiplier bit for bit instead of
 * "within the bisection tolerance 1e-6".  The parity tests pass the bracket of the eigenvalues they hand the oracle. */
int qcqpmi_admm_set_bracket(qcqpmi_ctx *ctx, const double *slo, const double *ehi);
/* out[k] = P_k Vin_k for every constraint k (n x p blocks, row-major; shared != 0: one Vin for all): the two passes over
 * the resident dense constraint matrices that a randomised range finder needs (Y = P Omega, Z = P Q) -- the device-side
 * part of building the reduced bases above without an O(n^3) eigendecomposition per constraint. */
int qcqpmi_admm_apply_constraints(qcqpmi_ctx *ctx, int p, const double *Vin, int shared, double *out);
/* onecons_qcqp(z, f_k) (utilities.py:149-196) for every resident point z, k = 1..m, with the installed eigenpairs /
 * basis: out = R x n projections (host layout of qcqpmi_pop_download); the population is not changed.  (The reference's
 * own test examples/tests/one_constraint_qcqp.py exercises exactly this function.) */
int qcqpmi_admm_onecons(qcqpmi_ctx *ctx, int64_t k, double *out);
/* Device-side setup of the z-update for a non-diagonal P0: (2 (P0 + rho m I))^-1 -- the matrix the reference factorises
 * with SuperLU (qcqp.py:224-227) -- by a Newton-Schulz iteration on the engine's own GEMM; kept packed inside the context.
 * A following qcqpmi_admm_run with the same rho may then pass Minv = NULL.  resid_out: estimate of max |I - M X| reached;
 * iters_out: iterations taken.  Fails (QCQPMI_EREFERENCE) if P0 + rho m I is not positive definite. */
int qcqpmi_admm_zsolver_device(qcqpmi_ctx *ctx, double rho, int64_t max_iter, double *resid_out, int64_t *iters_out);
/* lambda_min(P0) for the rho check / auto-rho of improve_admm (qcqp.py:262, 272: LA.eigh in the reference): Lanczos with
 * full reorthogonalisation, the products P0 v on the device.  max_steps <= 0: up to n steps. */
int qcqpmi_p0_lambda_min(qcqpmi_ctx *ctx, int64_t max_steps, double tol, double *lmin_out, int64_t *steps_out);
int qcqpmi_admm_run(qcqpmi_ctx *ctx, int phase1, int64_t num_iters, double tol, double viol_lim,
                    double rho, const double *Minv, int64_t *iters1, int64_t *iters2, double *f0,
                    double *maxviol);   /* Minv may be NULL if P0 is diagonal or after qcqpmi_admm_zsolver_device(rho) */
/* With reduced bases (qcqpmi_admm_set_basis, rp <= 8) and a diagonal P0 -- BASELINE.json configs[3] -- qcqpmi_admm_run runs
 * improve_admm (qcqp.py:254-285: phase 1, better, phase 2 with its bestx bookkeeping, better) inside ONE persistent kernel
 * per tile of 16 restarts (csrc/admm_fused.hip; a tile may be shared by a cluster of 2..16 workgroups so that few restarts
 * still fill the chip), no host in the loop.  qcqpmi_admm_fused(ctx, 0) selects the multi-launch path everywhere (the
 * cross-check; same iteration, other summation order in the two products); default 1.  qcqpmi_last_admm_kernel names the
 * path the last run took ("admm_fused_kernel" / "admm_multi_launch") and the workgroups per tile it used.
 * enable == 2 (round 5, experiment kept as a cross-check): the same kernel with FOUR-wave workgroups, two per compute unit,
 * twice as many workgroups per tile -- the co-resident workgroups belong to different tiles; measured equal to the
 * eight-wave geometry at 1024 restarts and slower elsewhere, hence not the default. */
int qcqpmi_admm_fused(qcqpmi_ctx *ctx, int enable);

/* Unit bases (round 5).  When every basis vector handed to qcqpmi_admm_set_basis is +-e_i -- separable constraints
 * p x_i^2 + q x_i + r ~ 0 (Boolean least squares, MAXCUT, boxes; /root/reference/examples/boolean_least_squares.py:34-36,
 * maxcut.py:25-28), for which the eigenvectors utilities.py:160-162 takes from LAPACK are unit vectors -- qcqpmi_admm_run
 * replaces the two consensus products of an iteration (qcqp.py:204-207, 236-239 in the basis) by a gather and a scatter:
 * the same values, no n x m operator.  With one row per constraint the z-update, the gather, the projections
 * (onecons_qcqp, utilities.py:149-196) and the scatter of an iteration are ONE launch (admm_unit_step_kernel; the same bits
 * as the separate launches).  On by default; enable = 0 forces the GEMM path (cross-check). */
int qcqpmi_admm_unit_bases(qcqpmi_ctx *ctx, int enable);
const char *qcqpmi_last_admm_kernel(qcqpmi_ctx *ctx, int *workgroups_per_tile);

/* Y = (sum_k w_k P_k) X for the resident population (w: m+1 weights, objective first; Y: R x n like
 * qcqpmi_pop_download).  Building block of the general SDP-relaxation solver (qcqp_amd/sdr.py: the gradient of
 * the Burer-Monteiro augmented Lagrangian is 2 S V with S = C + sum_k y_k M_k); one streaming pass over all
 * matrices + one GEMM.  For problems whose constraints are separable only the objective has a matrix: Y = w_0 P0 X
 * (the constraints are elementwise operators the caller applies itself). */
int qcqpmi_pop_weighted_product(qcqpmi_ctx *ctx, const double *w, double *Y);
/* S = sum_k w_k P_k itself (n x n row-major on the host): the dual matrix of the SDP certificate when the matrices
 * only exist on the device.  qcqpmi_get_linear returns q_k, r_k, relop of function k as the context holds them
 * (also for device-generated functions). */
int qcqpmi_weighted_matrix(qcqpmi_ctx *ctx, const double *w, double *S);
int qcqpmi_get_linear(qcqpmi_ctx *ctx, int64_t k, double *q, double *r, int *relop);
/* Quadratic and linear parts of every function for the resident population, kept apart:
 * quad[k][r] = x_r' P_k x_r + r_k,  lin[k][r] = q_k' x_r   ((m+1) x R each, row-major).  Same kernels as the
 * evaluation of QuadraticFunction.eval (utilities.py:49-50); the SDP solver needs the parts of the homogeneous forms. */
int qcqpmi_pop_eval_parts(qcqpmi_ctx *ctx, double *quad, double *lin);

/* solve_sdr (qcqp.py:72-97) for the UNIT-DIAGONAL family -- constraints x_i^2 = d_i, i.e. Boolean least
 * squares, MAXCUT, partitioning; the host scales d to 1:
 *     minimise <C, X>  s.t.  X_ii = 1, X PSD,   C symmetric N x N (N = n + 1, homogenised, row-major)
 * in Burer-Monteiro form X = V V^T (V: N x 64, unit rows) by the mixing method on the device.  V holds the
 * start on entry and the solution on return; hist[0] = start objective, hist[t] = objective after sweep
 * t, hist[*sweeps_done + 1] = objective recomputed at the end (hist has max_sweeps + 2 entries).
 * The reference delegates this to cvxpy + an SDP solver: parity unpinned, validated by optimality
 * conditions. */
int qcqpmi_sdr_solve_unitdiag(qcqpmi_ctx *ctx, const double *C, int64_t N, double *V, int max_sweeps,
                              double tol, double *hist, int *sweeps_done);

/* ---- QCQPForm.better ordering over the population (utilities.py:135-146):
 * lexicographic minimum of (int(maxviol/tol), f0), ties -> lowest index.  Evaluates the
 * population if needed.  best_x (n doubles) may be NULL. */
int qcqpmi_select_best(qcqpmi_ctx *ctx, double tol, int64_t *best_index, double *best_f0,
                       double *best_maxviol, double *best_x);

/* ---- timing of the hot kernels (HIP events on the context's stream) ---------------------
 * which: 0 = eval, 1 = cd phase 1, 2 = cd phase 2, 3 = sdr sampling, 4 = admm secular kernel.  Returns the duration of
 * the most recent launch of that kernel in milliseconds. */
int qcqpmi_last_kernel_ms(qcqpmi_ctx *ctx, int which, double *ms);
/* name of the phase-2 kernel the most recent qcqpmi_cd_run dispatched to: "cd_phase2_q_kernel" (pipelined, Boolean family),
 * "cd_phase2_rs_kernel" (its predecessor: box families, n not a multiple of 16, n > 1024), "cd_phase2_kernel" (general
 * separable constraints), "dense_chain_kernel" (constraints that couple coordinates, n > 64 or generated),
 * "cd_general_kernel" (coupled constraints in the reference's arithmetic); static storage.  The parity tests assert it. */
const char *qcqpmi_last_cd_kernel(qcqpmi_ctx *ctx);
/* Scheduling of phase 2 for the Boolean family (the headline kernel).  0: a workgroup is bound to a tile of 16 restarts for
 * the whole launch (cd_phase2_q_kernel: it runs until its slowest restart has converged).  1: a workgroup owns 16 SLOTS; a
 * restart that is done (qcqp.py:172-176, or num_iters sweeps) is written out at the next sweep boundary and its slot takes the
 * next restart from a device-side queue (cd_phase2_qs_kernel, csrc/cd_queue.hip).  Per restart the same arithmetic; results
 * do not depend on the scheduling (every product is summed in one association).  qcqpmi_last_cd_kernel names the kernel. */
int qcqpmi_cd_queue(qcqpmi_ctx *ctx, int mode);     /* 0 off, 1 on wherever it applies, 2 (default) auto: more tiles than CUs */
/* (ABI 4 removed the round-3 experiments around that kernel -- qcqpmi_cd_chain, qcqpmi_cd_partition, qcqpmi_cd_ring_start /
 * submit / collect / stop: launches that ran restarts of other contexts' populations, a CU-masked stream, one persistent
 * spin-waiting launch for several contexts.  They needed GPU_MAX_HW_QUEUES > 4, worked on the 192-CU partition only and could
 * stall; qcqpmi_cd_stream_run below does what they were after inside ONE self-contained launch.) */
/* debug (after qcqpmi_debug_profile enabled profiling): tick sums (s_memtime) over the workgroups of the last
 * qcqpmi_cd_stream_run launch, TWENTY-FOUR entries (ABI 5; 8 before) -- [0] column build (suggest + phase 1 + gate), [1] whole launch,
 * [2] episodes, [3] columns built, [4] the normals' share of [0], [5] the roles of the episodes (the rest: write-out, queue,
 * refill); cd_life_kernel only: [6] workgroups whose waves covered the four SIMDs evenly, [8] the chain waves' wait for
 * partial tiles, [9] / [10] multiplying wave 0's waits for a commit / for its slot, [11] block intervals, [12] blocks with a
 * near-tie replay, [13] / [14] / [15] / [7] the chain's stages: sum of the partial tiles + requests, the 16 steps, block end +
 * commit, fix-up + own share + staging, [16] the longest workgroup's [1] (ticks of the launch as the device saw it: with the
 * HIP-event duration the tick rate; [1] / workgroups / [16] = how much of the launch the average workgroup was alive), [17] column
 * build up to the end of phase 1, [18] the chain's prologue per episode (virtual interval, operands of block 0), [19] issuing an
 * interval's memory requests (top of the chain's loop: not part of [13]), [20] from the chain's last interval to the barrier
 * that ends the episode */
int qcqpmi_debug_life_profile(qcqpmi_ctx *ctx, int64_t *out24);
/* POPULATION STREAMING (round 4) -- the reference's user loop `for ...: suggest(); improve(COORD_DESCENT)` (README.md:51-57)
 * for K populations of R restarts in ONE persistent launch: a workgroup owns 16 restart slots; a slot that becomes free draws
 * the next restart index of the run and runs that restart's WHOLE step itself -- suggest(RANDOM) (qcqp.py:381-382; the keyed
 * normals of qcqpmi_pop_randn), phase 1 (qcqp.py:101-149), the gate (qcqp.py:189), phase 2 to convergence (qcqp.py:152-178),
 * objective and max violation of the result -- so the matrix pipes work on live restarts across population boundaries and no
 * preparation kernel, second stream or CU partition exists (csrc/cd_queue.hip, lifecycle mode).  Population p uses the seed
 * seed + p seed_stride and the global restart indices first_index + p first_stride + [0, R): every restart equals the one
 * qcqpmi_pop_randn(R, seed_p, first_p) + qcqpmi_cd_run would produce (same draws, same moves; the reported objective is a fresh
 * evaluation of the final point summed from the products of the restart's last sweep instead of the tracked value: 1e-12
 * relative).  generate = 0: the K R resident points (qcqpmi_pop_upload) are the starts instead of normals.  Outputs: per-restart
 * arrays of K R entries (population-major; may be NULL) as in qcqpmi_cd_run, and per population the best restart (index within
 * the population, QCQPForm.better ordering with bucket width select_tol), its objective, max violation and point (K x n).
 * The K R final points stay resident (qcqpmi_pop_download).
 * Round 5 (ABI 5): the launch is cd_life_kernel (csrc/cd_life.hip) -- four-wave workgroups, two per CU, the X tile in a private
 * global tile instead of LDS -- and takes every problem with separable constraints of ONE class, one constraint per coordinate
 * (Boolean / MAXCUT x_i^2 == 1, box and disc x_i^2 <= c, one-sided and linear single-coordinate constraints), a diagonal of P0
 * that is positive everywhere or ZERO everywhere (MAXCUT: qcqp.py:152-178 with a linear scalar objective), any 48 <= n <= 2304
 * (n need not be a multiple of 16; 1024 < n <= 2304 runs eight-wave workgroups).  Round 6 (ABI 6): ALSO problems with up to four
 * classes of coordinates (coordinates whose constraint lists are bit-identical share a class) and up to two constraints per
 * coordinate -- boxes with different bounds, an annulus beside an equality, two linear bounds, MAXCUT with relaxed vertices
 * (qcqp.py:113-141, 160-176 treat every coordinate's list on its own); ONE lifecycle kernel for every shape; with
 * qcqpmi_cd_set_objective_factor the kernel carries L^T X instead of multiplying with P0 and takes n up to 4096.
 * QCQPMI_EUNSUPPORTED otherwise (more than four classes, three or four constraints per coordinate, mixed diagonal signs, coupled
 * constraints, n > 2304 without an objective factor): use
 * qcqpmi_cd_run per population; a refused call leaves the resident population untouched.  K R < 2^30 (restart tickets are 32-bit; QCQPMI_EINVAL beyond).
 * Near-ties: a restart whose decision is within rounding of a tie is replayed in the reference's arithmetic like in
 * qcqpmi_cd_run; for a positive diagonal the replay sees the objective RELATIVE to the start of phase 2 (the constant of its
 * scalar objective differs from the reference's by f0 at that start: candidates closer than one ulp of f0 may resolve
 * differently; exact ties do not), for a zero diagonal the absolute objective (evaluated by an extra frozen sweep). */
int qcqpmi_cd_stream_run(qcqpmi_ctx *ctx, int64_t K, int64_t R, int generate, int phase1, int64_t num_iters, double viol_tol,
                         double tol, uint64_t seed, uint64_t seed_stride, uint64_t first_index, uint64_t first_stride,
                         double select_tol, int64_t *sweeps1, int64_t *sweeps2, int64_t *visits2, int64_t *accepted2,
                         uint8_t *ran_phase2, double *f0, double *maxviol, int64_t *best_index, double *best_f0,
                         double *best_maxviol, double *best_x);
/* Device and pinned-host buffers for a qcqpmi_cd_stream_run(K, R) to come (population, per-restart outputs, per-population
 * winners): allocation only, so that a timed or latency-sensitive run does not start with hipMalloc / hipHostMalloc.  A
 * resident population smaller than K R points is dropped (like any reallocation of the population). */
int qcqpmi_cd_stream_reserve(qcqpmi_ctx *ctx, int64_t K, int64_t R);
/* Which lifecycle kernel qcqpmi_cd_stream_run launches: 0 (default) the faster one for the shape -- cd_life_kernel (csrc/cd_life.hip,
 * round 5) everywhere except the Boolean family at n >= 960 with more than 8192 restarts in the run, where the round-4 kernel
 * (cd_phase2_qs_kernel<lifecycle>, csrc/cd_queue.hip: Boolean family, n a multiple of 16, n <= 1024) is 12 % faster; 2: cd_life_kernel
 * wherever it applies; 1: the round-4 kernel only.  Both produce the same restarts (tests/test_gpu_life.py, tests/test_gpu_stream.py run
 * every case with both); qcqpmi_last_cd_kernel names the one that ran. */
int qcqpmi_cd_life_version(qcqpmi_ctx *ctx, int version);
/* Factored objective (round 6).  get_onevar_func (utilities.py:99-105) needs (P0 z)_i per coordinate visit; when P0 = L L^T with L
 * n x r (row-major; a least-squares objective |A x - b|^2 has L = A^T, r = rows of A; any PSD P0 of low rank has one), the lifecycle
 * kernel carries Y = L^T X per tile of restarts instead of multiplying with P0: (P0 X)[I_b, :] = L[I_b, :] Y and Y += L[I_b, :]^T (moves
 * of block b) -- 8 r / 16 matrix instructions per block of 16 coordinates instead of n / 4.  P0 itself stays uploaded (diagonal
 * blocks, the evaluation kernels, every other path); the caller vouches for L L^T = P0 (qcqp_amd.lowrank.objective_factor checks
 * it to 1e-12 of the largest entry).  Takes r <= 288 and 128 <= n <= 4096 (ABI 6), single-class separable constraints on a positive
 * diagonal (QCQPMI_EUNSUPPORTED otherwise; qcqpmi_cd_stream_run then runs as without a factor).  L == NULL or r == 0 removes it;
 * qcqpmi_cd_life_version(ctx, 3) = cd_life_kernel WITHOUT the factor (comparisons). */
int qcqpmi_cd_set_objective_factor(qcqpmi_ctx *ctx, const double *L, int64_t r);
/* Coordinate descent for constraints that couple coordinates IN THE REFERENCE'S SUMMATION ORDER (test / diagnostic mode, any
 * n): every one-variable coefficient (t2, t1, t0) of get_onevar_func (utilities.py:99-105) is formed by row-sequential sums
 * like the reference's CSR products -- t0 = f_k(z) afresh per coordinate, O((m+1) n^2) per coordinate visit -- so that
 * trajectories are comparable with the reference value for value where the MFMA path (different summation order, 1e-13
 * input noise amplified by the phase-1 bisection) is only comparable as a distribution.  Needs uploaded (not generated)
 * functions with m n^2 <= 2e9 entries.  enable = 0 restores the default dispatch. */
int qcqpmi_cd_reference_order(qcqpmi_ctx *ctx, int enable);
/* The UNIT STEP of the default dense-constraint path (products on the matrix cores + dense_chain_mw_kernel) on the resident
 * points: the coordinate visits [coord_lo, coord_hi) (0, 16 = all) of block `block` in sweep `sweep` of phase 1
 * (qcqp.py:112-141) or phase 2 (qcqp.py:160-176), started from fresh function values, with the keyed draws of that sweep.
 * phase 2: slack (R values, host) is the `viol` the reference fixes at the start of the phase (qcqp.py:157); NULL takes the
 * max violation of the resident points.  This is what the parity tests teacher-force with the states the oracle visits:
 * coordinate descent with coupled constraints is chaotic in the reference itself (one ulp of x0 moves 40 % of the dense
 * family's and every beamforming restart by > 1e-6: profiles/r04_reference_sensitivity.md, tests/test_host_cpu.py), so a
 * free-running trajectory can only be compared for bit-identical arithmetic (qcqpmi_cd_reference_order) -- the fast path is
 * compared visit by visit and block by block on the reference's own states. */
int qcqpmi_cd_dense_block_step(qcqpmi_ctx *ctx, int phase, int64_t sweep, int64_t block, int coord_lo, int coord_hi,
                               double viol_tol, double tol, uint64_t seed, uint64_t first_index, const double *slack);
/* Chain kernel of the dense-constraint path: 0 (default) dense_chain_mw_kernel -- a workgroup of four waves per restart,
 * the functions dealt to 256 threads -- whenever m + 1 <= 2048; 1: dense_chain_kernel, one wave per restart (round 1-2
 * kernel, the cross-check: the two produce the same points bit for bit). */
int qcqpmi_dense_chain_mode(qcqpmi_ctx *ctx, int mode);
/* Geometry of dense_chain_mw_kernel for m constraints (host-only, no device call): out4 = {slots per thread, threads that
 * hold constraints Tc, index of the serial thread (holds the objective), threads per restart}.  Constraint k = 1..m is
 * slot (k - 1) / Tc of thread (k - 1) % Tc.  More than 8 slots: the one-wave kernel is used instead. */
int qcqpmi_dense_chain_geometry(int64_t m, int *out4);
int qcqpmi_sync(qcqpmi_ctx *ctx);
/* debug: enable in-kernel cycle counters of the phase-2 kernel / read their sums over tiles
 * (slots: 0 mfma, 1 feasible sets, 2 barrier, 3 sequential part, 4 barrier, 5 #blocks) */
int qcqpmi_debug_profile(qcqpmi_ctx *ctx, int enable, int64_t *sums8);
/* debug: stage cycle counters (s_memtime ticks of member 0 of tile 0) of the last fused ADMM run made while
 * qcqpmi_debug_profile(ctx, 1, NULL) was on: 0 z-update, 1 partial product, 2 exchange 1, 3 sums, 4 secular solves,
 * 5 exchange 2, 6 gather, 7 bookkeeping, 9 iterations */
int qcqpmi_debug_admm_profile(qcqpmi_ctx *ctx, int64_t *out16);
/* debug: stage tick sums (s_memtime) of the dense chain kernel's wave of restart 0 over the runs made while
 * qcqpmi_debug_profile(ctx, 1, NULL) was on (library built with -DDN_PROFILE=1, else zeros): [0..15] phase 1, [16..31]
 * phase 2 of the serial thread (one-wave kernel: the wave), [32..63] the same for thread 0 of the multi-wave kernel.
 * Slots: 0 set-up, 1 one-variable coefficients, 2 bounds + reductions, 3 gaps, 4 segment sweep, 5 minimiser / draw,
 * 6 commit, 7 write-back, 8 coordinates visited, 9 feasible-set evaluations.  Reading resets the sums. */
int qcqpmi_debug_dense_profile(qcqpmi_ctx *ctx, int64_t *out64);
/* debug: per-wave event trace (cycle stamps) of tile 0 of the last profiled phase-2 run; count <= 2048 words */
int qcqpmi_debug_trace(qcqpmi_ctx *ctx, int64_t *out, int count);

/* ---- multi-GPU: one process per GPU, restarts sharded by global index, ONE collective at the
 * end to pick the global best (RCCL over xGMI; librccl is loaded lazily). --------------- */
int qcqpmi_comm_unique_id(uint8_t id_out[128]);
int qcqpmi_comm_init(qcqpmi_ctx *ctx, int rank, int world, const uint8_t id[128]);
/* all ranks call; local best (from qcqpmi_select_best semantics) -> global best on every rank.
 * index_offset = global index of this rank's candidate 0. */
int qcqpmi_comm_select_best(qcqpmi_ctx *ctx, double tol, int64_t index_offset,
                            int64_t *best_global_index, double *best_f0, double *best_maxviol,
                            double *best_x);
int qcqpmi_comm_barrier(qcqpmi_ctx *ctx);
/* in-place all-reduce of host doubles (RCCL all-reduce of a device copy); op 0 = max, 1 = sum (bench.py: max-over-ranks
 * step time, sum-over-ranks work counters; the exchange of a streamed run: the table of the local winners' keys of all
 * populations, then the table of the winners' points -- qcqp_amd.dist.global_best_of_populations) */
int qcqpmi_comm_allreduce(qcqpmi_ctx *ctx, double *values, int64_t count, int op);
/* all-gather of host bytes (ABI 5): every rank contributes nbytes bytes, recv receives world x nbytes bytes in rank order -- one
 * ncclAllGather.  The exchange of a streamed run uses it for the local winners' KEYS: 32 bytes per population and rank -- the
 * QCQPForm.better bucket int(maxviol / tol) (utilities.py:139-140), the objective, the max violation, the GLOBAL restart index
 * (int64: exact; the ticket cap K R < 2^30 of qcqpmi_cd_stream_run bounds it per rank, not per job) -- then ONE sum-all-reduce of
 * the K winners' points with only the owner's rows filled (qcqp_amd.dist.global_best_of_populations). */
int qcqpmi_comm_allgather(qcqpmi_ctx *ctx, const void *send, int64_t nbytes, void *recv);

#ifdef __cplusplus
}
#endif
#endif
