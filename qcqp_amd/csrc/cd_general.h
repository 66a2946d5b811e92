// Coordinate descent for constraints that COUPLE coordinates (dense / general quadratic constraints,
// e.g. the beamforming family): first correct GPU path, not yet a fast one.
//
//   coord_descent_phase1   qcqp.py:101-148        coord_descent_phase2   qcqp.py:152-178
//   get_onevar_func        utilities.py:99-105    onevar_qcqp            utilities.py:241-288
//
// One workgroup (256 threads) owns a tile of 16 restarts, thread = (function slot s, restart r).
// For coordinate i:
//   A. every thread computes, for the functions k = s, s+16, ... (k = 0 objective, 1..m constraints),
//      the one-variable coefficients (t2, t1, t0) of f_k in x_i -- a row dot P_k[i,:] . x summed in
//      the reference's order (bit-identical t1), t0 from an incrementally tracked f_k(x);
//   B. one thread per restart intersects the feasible intervals of the constraints that involve x_i
//      (running list of disjoint segments), applies the reference's end-point rules (segments that
//      end at +inf, zero-width segments and segments whose right end is shared by two constraints
//      are dropped), minimises the scalar objective (phase 2) or bisects on the
//      achievable slack (phase 1), and commits the move.
// The heavy part (A) is a (m+1) x n by n x 16 product per coordinate and belongs on the matrix cores
// (blocked like cd_phase2_kernel); B wants a wave-cooperative intersection.  Both are next steps.
#pragma once
#include "kernels.h"
#include "onevar.h"

namespace qcqpmi {

constexpr int GEN_CAP = 48;   // capacity of a restart's segment list

struct CdGenArgs {
    CdArgs b;
    const double *gP;     // [m][n][n] dense constraint matrices (row-major)
    const double *F;      // (m+1) x Rpad: f_k(x) at entry (row 0: objective), from eval_kernel
    int64_t Rpad;
    int exact_t0;         // 1: t0 = f_k(z) evaluated afresh in the reference's summation order (small n);
                          // 0: t0 from the incrementally tracked f_k(x) (rounding-level different)
};

// get_onevar_func (utilities.py:99-105) of function k (0: the objective) in coordinate i for the restart in column r of the
// tile X ([n16][16]); gP: dense constraint matrices [m][n][n].  t2 = P[i, i], t1 = 2 P[i, :] z + q[i] with z = x, z[i] = 0.
__device__ inline void gen_onevar_t2t1(const DevProblem &P, const double *gP, const double *X, int r, int64_t i, int k,
                                       double *t2, double *t1) {
    const int64_t n = P.n;
    const double *row = (k == 0) ? (P.P0 + i * P.n16) : (gP + ((int64_t)(k - 1) * n + i) * n);
    double d = 0.0;
    for (int64_t j = 0; j < n; j++)
        if (j != i) d += row[j] * X[j * 16 + r];
    const double qk = (k == 0) ? P.q0[i] : P.gq[(int64_t)(k - 1) * P.n16 + i];
    *t2 = row[i];
    *t1 = 2.0 * d + qk;
}

// t0 = (P.dot(z) + q).dot(z) + r with z = x, z[i] = 0, rows and columns in index order (utilities.py:104)
__device__ inline double gen_onevar_t0(const DevProblem &P, const double *gP, const double *X, int r, int64_t i, int k) {
    const int64_t n = P.n;
    const double *Pk = (k == 0) ? P.P0 : (gP + (int64_t)(k - 1) * n * n);
    const int64_t ld = (k == 0) ? P.n16 : n;
    const double *qv = (k == 0) ? P.q0 : (P.gq + (int64_t)(k - 1) * P.n16);
    double acc = 0.0;
    for (int64_t i2 = 0; i2 < n; i2++) {
        double rw = 0.0;
        for (int64_t j = 0; j < n; j++)
            if (j != i) rw += Pk[i2 * ld + j] * X[j * 16 + r];
        const double z2 = (i2 == i) ? 0.0 : X[i2 * 16 + r];
        acc += (rw + qv[i2]) * z2;
    }
    return acc + ((k == 0) ? P.r0 : P.gr[k - 1]);
}

struct SegList {
    double *lo, *hi;
    int *cnt;
    int n;
};

// C = feasible set of the coordinate at slack s: running intersection over the involved constraints
// (coef: [(m+1)][16][3] in LDS).  Returns the number of involved constraints (mf) through *mf_out
// and the final list in L (filtered with the reference's end-point rules).
__device__ inline void general_feasible_set(const double *coef, const int *grel, int m, int r, double s,
                                            SegList &A, SegList &B, int *mf_out, int *overflow) {
    A.lo[0] = -QM_INF; A.hi[0] = QM_INF; A.cnt[0] = 1;   // the base interval (-inf, +inf)
    A.n = 1;
    int mf = 0;
    for (int k = 1; k <= m; k++) {
        const double t2 = coef[(k * 16 + r) * 3], t1 = coef[(k * 16 + r) * 3 + 1], t0 = coef[(k * 16 + r) * 3 + 2];
        if (t2 == 0.0 && t1 == 0.0) continue;   // qcqp.py:116,166
        mf++;
        const Seg2 iv = feasible_intervals(t2, t1, t0, grel[k - 1], s);
        B.n = 0;
        for (int a = 0; a < A.n; a++) {
            const double slo = A.lo[a], shi = A.hi[a];
            const int sc = A.cnt[a];
            for (int j = 0; j < iv.n; j++) {
                const double il = j == 0 ? iv.lo0 : iv.lo1, ih = j == 0 ? iv.hi0 : iv.hi1;
                const double l = slo > il ? slo : il, h = shi < ih ? shi : ih;
                if (l <= h) {
                    if (B.n >= GEN_CAP) { *overflow = 1; continue; }
                    B.lo[B.n] = l; B.hi[B.n] = h;
                    B.cnt[B.n] = (ih < shi) ? 1 : ((ih == shi) ? sc + 1 : sc);   // #intervals ending exactly at h
                    B.n++;
                }
            }
        }
        SegList t = A; A = B; B = t;
    }
    // end-point rules of the counting sweep
    int w = 0;
    for (int a = 0; a < A.n; a++) {
        const bool keep = A.lo[a] != A.hi[a] && A.cnt[a] == 1;
        if (keep) { A.lo[w] = A.lo[a]; A.hi[w] = A.hi[a]; A.cnt[w] = 1; w++; }
    }
    A.n = w;
    *mf_out = mf;
}

// scalar minimiser over a segment list (utilities.py:257-288); 1 + *xout, 0 for None, < 0 where the
// reference raises
__device__ inline int general_minimise(double p0, double q0, double r0, const SegList &C, const DrawKey &dk,
                                       double *xout) {
    if (C.n == 0) return 0;
    if (p0 == 0.0 && q0 == 0.0) {
        U4 rnd = cd_draw(dk.seed, dk.restart, dk.coord, dk.sweep_tag, dk.iter);
        const int c = draw_choice(rnd, C.n);
        const double lo = C.lo[c], hi = C.hi[c];
        if (__builtin_isinf(lo) || __builtin_isinf(hi)) return -1;
        *xout = draw_uniform(rnd, lo, hi);
        return 1;
    }
    const double x0 = (p0 > 0.0) ? -q0 / (2.0 * p0) : QM_NAN;
    double bestf = QM_INF;
    int nb = 0, err = 0;
    for (int j = 0; j < C.n; j++) {
        if (C.lo[j] <= x0 && x0 <= C.hi[j]) { *xout = x0; return 1; }
        const double fl = onevar_eval(p0, q0, r0, C.lo[j], &err);
        const double fr = onevar_eval(p0, q0, r0, C.hi[j], &err);
        if (bestf > fl) { nb = 1; bestf = fl; } else if (bestf == fl) nb++;
        if (bestf > fr) { nb = 1; bestf = fr; } else if (bestf == fr) nb++;
    }
    if (err) return -2;
    if (nb == 0) return 0;
    int idx = 0, seen = 0;
    if (nb > 1) {
        U4 rnd = cd_draw(dk.seed, dk.restart, dk.coord, dk.sweep_tag, dk.iter);
        idx = draw_choice(rnd, nb);
    }
    for (int j = 0; j < C.n; j++) {
        int e2 = 0;
        const double fl = onevar_eval(p0, q0, r0, C.lo[j], &e2), fr = onevar_eval(p0, q0, r0, C.hi[j], &e2);
        if (fl == bestf) { if (seen == idx) { *xout = C.lo[j]; return 1; } seen++; }
        if (fr == bestf) { if (seen == idx) { *xout = C.hi[j]; return 1; } seen++; }
    }
    return 0;
}

template <int PHASE>
__global__ __launch_bounds__(256) void cd_general_kernel(CdGenArgs ga) {
    extern __shared__ double smem[];
    const CdArgs &a = ga.b;
    const DevProblem &P = a.P;
    const int tid = threadIdx.x, r = tid & 15, slot = tid >> 4;
    const int64_t tile = blockIdx.x;
    const int64_t n = P.n, n16 = P.n16;
    const int m = (int)P.m;
    double *X = a.X + tile * n16 * 16;
    const int64_t gr = tile * 16 + r;
    // ---- LDS
    double *sp = smem;
    double *coef = sp; sp += (size_t)(m + 1) * 16 * 3;
    double *Fk = sp; sp += (size_t)(m + 1) * 16;
    double *llo = sp; sp += 2 * 16 * GEN_CAP;
    double *lhi = sp; sp += 2 * 16 * GEN_CAP;
    int *lcnt = (int *)sp; sp += 16 * GEN_CAP;   // 2 * 16 * CAP ints
    int *ctrl = (int *)sp;                        // [0] live restarts, [1] any restart wants this sweep

    // f_k(x) of every function in the reference's own summation order, (P.dot(x) + q).dot(x) + r (utilities.py:49-50): rows and
    // columns in index order -- bit-identical to the reference / oracle.  Used in the reference-order mode wherever the
    // reference evaluates afresh: the violations after a phase-1 sweep (qcqp.py:142) and the slack of phase 2 (qcqp.py:157).
    auto fresh_F = [&]() {
        for (int k = slot; k <= m; k += 16) {
            const double *Pk = (k == 0) ? P.P0 : (ga.gP + (int64_t)(k - 1) * n * n);
            const int64_t ld = (k == 0) ? n16 : n;
            const double *qv = (k == 0) ? P.q0 : (P.gq + (int64_t)(k - 1) * n16);
            double acc = 0.0;
            for (int64_t i2 = 0; i2 < n; i2++) {
                double rw = 0.0;
                for (int64_t j = 0; j < n; j++) rw += Pk[i2 * ld + j] * X[j * 16 + r];
                acc += (rw + qv[i2]) * X[i2 * 16 + r];
            }
            Fk[k * 16 + r] = acc + ((k == 0) ? P.r0 : P.gr[k - 1]);
        }
    };
    if (ga.exact_t0) fresh_F();
    else for (int k = slot; k <= m; k += 16) Fk[k * 16 + r] = (gr < a.R) ? ga.F[(int64_t)k * ga.Rpad + gr] : 0.0;
    __syncthreads();
    // per-restart state, held by the slot-0 thread of the restart
    const bool owner = slot == 0;
    bool live = owner && gr < a.R && (PHASE == 1 || a.flag[gr]);
    double slack = (PHASE == 2 && owner && gr < a.R) ? a.slack[gr] : 0.0;
    if (PHASE == 2 && owner && ga.exact_t0) {      // max(prob.violations(x)) afresh (qcqp.py:157), not the evaluation kernel's
        double v = -QM_INF;
        for (int k = 1; k <= m; k++) {
            const double f = Fk[k * 16 + r];
            const double w = (P.grel[k - 1] == RELOP_EQ) ? fabs(f) : (f > 0.0 ? f : 0.0);
            v = w > v ? w : v;
        }
        slack = v;
    }
    int64_t upd_counter = 0, visits = 0, accepted = 0, sweeps = 0;
    double viol_last = QM_INF;
    int status = 0, overflow = 0;
    bool sweep_on = false;   // this restart takes part in the current sweep
    __syncthreads();

    for (int64_t t = 0; t < a.num_iters; t++) {
        // loop-top tests of the reference
        if (owner) {
            if (PHASE == 1 && live && viol_last < a.viol_tol) live = false;   // qcqp.py:111
            sweep_on = live;
            if (sweep_on) sweeps++;
        }
        if (tid == 0) { ctrl[0] = 0; }
        __syncthreads();
        if (owner && live) atomicAdd(&ctrl[0], 1);
        __syncthreads();
        if (ctrl[0] == 0) break;
        for (int64_t i = 0; i < n; i++) {
            // ---- A. one-variable coefficients of every function in x_i (utilities.py:99-105)
            const double xi = X[i * 16 + r];
            for (int k = slot; k <= m; k += 16) {
                double t2, t1;
                gen_onevar_t2t1(P, ga.gP, X, r, i, k, &t2, &t1);
                // t0: f_k(z) afresh in the reference's summation order (utilities.py:104; bit-identical to the reference /
                // oracle, O(n^2) per function) or from the tracked f_k(x)
                const double t0 = ga.exact_t0 ? gen_onevar_t0(P, ga.gP, X, r, i, k) : Fk[k * 16 + r] - xi * (t2 * xi + t1);
                coef[(k * 16 + r) * 3] = t2; coef[(k * 16 + r) * 3 + 1] = t1; coef[(k * 16 + r) * 3 + 2] = t0;
            }
            __syncthreads();
            // ---- B. the move of restart r
            if (owner && sweep_on) {
                SegList A, B;
                A.lo = llo + r * GEN_CAP; A.hi = lhi + r * GEN_CAP; A.cnt = lcnt + r * GEN_CAP;
                B.lo = llo + (16 + r) * GEN_CAP; B.hi = lhi + (16 + r) * GEN_CAP; B.cnt = lcnt + (16 + r) * GEN_CAP;
                int mf = 0;
                bool moved = false;
                double xn = xi;
                visits++;
                if (PHASE == 2) {
                    general_feasible_set(coef, P.grel, m, r, slack, A, B, &mf, &overflow);
                    DrawKey dk{a.seed, a.first_index + (uint64_t)gr, (uint32_t)i, (uint32_t)t | 0x80000000u, 0u};
                    const int got = general_minimise(coef[r * 3], coef[r * 3 + 1], coef[r * 3 + 2], A, dk, &xn);
                    if (got < 0) { status = got; live = false; sweep_on = false; }
                    else if (got && fabs(xn - xi) > a.tol) { moved = true; upd_counter = 0; accepted++; }
                    else {
                        upd_counter++;
                        if (upd_counter == n) { live = false; sweep_on = false; }   // converged (qcqp.py:172-176)
                    }
                } else {
                    // local violation over the involved constraints (qcqp.py:117)
                    double viol = -QM_INF;
                    int involved = 0;
                    for (int k = 1; k <= m; k++) {
                        const double t2 = coef[(k * 16 + r) * 3], t1 = coef[(k * 16 + r) * 3 + 1], t0 = coef[(k * 16 + r) * 3 + 2];
                        if (t2 == 0.0 && t1 == 0.0) continue;
                        involved++;
                        const double f = xi * (t2 * xi + t1) + t0;
                        const double v = (P.grel[k - 1] == RELOP_EQ) ? fabs(f) : (f > 0.0 ? f : 0.0);
                        viol = v > viol ? v : viol;
                    }
                    if (involved == 0) { status = -3; live = false; sweep_on = false; }   // ValueError (qcqp.py:117)
                    else {
                        double new_viol = viol, ss = -a.tol, es = viol - a.viol_tol;
                        uint32_t it = 0;
                        while (es - ss > a.tol) {
                            const double s = (ss + es) / 2.0;
                            general_feasible_set(coef, P.grel, m, r, s, A, B, &mf, &overflow);
                            DrawKey dk{a.seed, a.first_index + (uint64_t)gr, (uint32_t)i, (uint32_t)t, it++};
                            double xc;
                            const int got = general_minimise(0.0, 0.0, 0.0, A, dk, &xc);
                            if (got < 0) { status = got; live = false; sweep_on = false; break; }
                            if (!got) ss = s;
                            else { xn = xc; new_viol = s; es = s; }
                        }
                        if (status == 0) {
                            if (new_viol < viol) { moved = true; upd_counter = 0; accepted++; }
                            else {
                                upd_counter++;
                                if (upd_counter == n) sweep_on = false;   // "failed": leaves this sweep only (qcqp.py:138-141)
                            }
                        }
                    }
                }
                if (moved) {
                    X[i * 16 + r] = xn;
                    const double dlt = xn - xi;
                    for (int k = 0; k <= m; k++)   // f_k(x) follows the move: delta (t2 (xn + xi) + t1)
                        Fk[k * 16 + r] += dlt * (coef[(k * 16 + r) * 3] * (xn + xi) + coef[(k * 16 + r) * 3 + 1]);
                }
            }
            __syncthreads();
        }
        if (PHASE == 1 && ga.exact_t0) {     // (uniform branch) the reference evaluates the violations afresh after the sweep
            __syncthreads();
            fresh_F();
            __syncthreads();
        }
        if (PHASE == 1 && owner && live) {
            // viol = max(prob.violations(x)) (qcqp.py:142) from the tracked f_k (reference-order mode: evaluated afresh above)
            double v = -QM_INF;
            for (int k = 1; k <= m; k++) {
                const double f = Fk[k * 16 + r];
                const double w = (P.grel[k - 1] == RELOP_EQ) ? fabs(f) : (f > 0.0 ? f : 0.0);
                v = w > v ? w : v;
            }
            viol_last = v;
        }
    }
    if (owner && gr < a.R) {
        a.visits[gr] = visits; a.accepted[gr] = accepted; a.sweeps[gr] = sweeps;
        a.status[gr] = overflow ? -4 : status;
        if (PHASE == 1) a.flag[gr] = (viol_last < a.viol_tol) ? 1 : 0;
    }
}

}  // namespace qcqpmi
