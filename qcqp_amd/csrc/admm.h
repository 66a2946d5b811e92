// Consensus ADMM improve (qcqp.py:195-285) for a whole population, in the eigenbasis of each
// constraint.
//
// The reference keeps, per restart, one copy x_k and one dual u_k of the n-vector for each of the
// m constraints and per iteration calls onecons_qcqp(z + u_k, f_k) (utilities.py:149-196): rotate
// into the eigenbasis of P_k (Q_k^T v), solve the secular equation by bisection, rotate back
// (Q_k xhat).  Here the state lives IN the eigenbasis: uh_k = Q_k^T u_k.  Per iteration
//     ZQ = [Q_1^T; ...; Q_m^T] Z                  one dgemm, (m n) x n by n x R
//     per (k, r): vhat = ZQ_k + uh_k; xhat_k = secular(vhat);  uh_k += ZQ_k - xhat_k
//                 D_k = xhat_k - uh_k                            admm_secular_kernel (hand-written)
//     S = sum_k x_k - sum_k u_k = [Q_1 ... Q_m] D   one dgemm, n x (m n) by (m n) x R
//     z = S / m   (phase 1)      z = (2 (P0 + rho m I))^-1 (2 rho S - q0)   (phase 2)
// which is algebraically the reference's iteration (Q_k orthogonal) without ever materialising
// x_k or u_k.  The two big products are PLAIN GEMMs and go to rocBLAS; everything with the
// algorithm's control flow (early exit of feasible inequality constraints, bracket, bisection to
// 1e-6 on the multiplier, violations, the `better` bookkeeping) is in the kernels below.
//
// Layout: column-major, one column per restart: Z is n x R (ld n), ZQ / UH are (m n) x R (ld m n),
// block k of a column is contiguous -> one wave owns one (constraint, restart) pair and streams its
// n-vector with fully coalesced loads; its share stays in registers across the bisection.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "onevar.h"

namespace qcqpmi {

struct AdmmArgs {
    int64_t n, m, R;
    const double *lam;    // [m][n] eigenvalues (numpy eigh order)
    const double *qhat;   // [m][n] Q_k^T q_k
    const double *rk;     // [m]
    const int *relop;     // [m]
    const double *slo;    // [m] bracket start  max_{lam>0} -1/lam  (or -inf)
    const double *ehi;    // [m] bracket end    min_{lam<0} -1/lam  (or +inf)
    double *ZQ;           // in: Q_k^T z ; out: D_k = xhat_k - uh_k(new)
    double *UH;           // in/out
    const uint8_t *act;   // [R] restart still iterating
    unsigned long long *mvbits;  // [R] max violation of z over the constraints, as ordered bits
    int first_iter;       // 1: xs = x0, us = 0: the duals are zero and UH is not read
    int viol_only;        // 1: only the violations of z are wanted (no update)
    double sec_tol;       // 1e-6 (utilities.py:149)
};

// wave-wide sum on DPP (row_shr prefix sums inside the rows of 16 lanes, row_bcast across them; lanes without
// a source add 0), total broadcast from lane 63 -- a shuffle butterfly costs 12 ds_bpermute per double
template <int CTRL, int ROW_MASK>
__device__ inline double admm_dpp0(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);      // +0.0 where there is no source lane
}

__device__ inline double wave_sum(double v) {
    v += admm_dpp0<0x111, 0xf>(v);   // row_shr:1
    v += admm_dpp0<0x112, 0xf>(v);   // row_shr:2
    v += admm_dpp0<0x114, 0xf>(v);   // row_shr:4
    v += admm_dpp0<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of every row holds the row's sum
    v += admm_dpp0<0x142, 0xa>(v);   // row_bcast:15 into rows 1, 3
    v += admm_dpp0<0x143, 0xc>(v);   // row_bcast:31 into rows 2, 3 -> lane 63 holds the total
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// num / den through the hardware reciprocal + two Newton steps (relative error ~1e-16; the exact IEEE division
// expands to ~30 instructions and the secular function costs 16 of them per lane and bisection step).  The
// multiplier is only located to 1e-6 (utilities.py:149), so last-bit differences of phi are immaterial.
__device__ inline double admm_div(double num, double den) {
    double r = __builtin_amdgcn_rcp(den);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    return num * r;
}

// one wave per (constraint k, restart r); EPL = elements per lane (n <= 64 EPL)
template <int EPL>
__global__ __launch_bounds__(256) void admm_secular_kernel(AdmmArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t widx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (widx >= a.m * a.R) return;
    const int64_t r = widx / a.m, k = widx % a.m;   // consecutive waves: consecutive constraints of a restart
    if (!a.act[r]) return;
    const int64_t n = a.n;
    double *zq = a.ZQ + (r * a.m + k) * n;
    double *uh = a.UH + (r * a.m + k) * n;
    const double *lm = a.lam + k * n, *qh = a.qhat + k * n;
    const double rk = a.rk[k];
    const int relop = a.relop[k];

    double L[EPL], Qh[EPL], V[EPL], Zq[EPL];
    double fz_a = 0.0, fz_b = 0.0, fv_a = 0.0, fv_b = 0.0;
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        const int64_t j = lane + 64 * e;
        const bool ok = j < n;
        L[e] = ok ? lm[j] : 0.0;
        Qh[e] = ok ? qh[j] : 0.0;
        Zq[e] = ok ? zq[j] : 0.0;
        const double u = (ok && !a.first_iter) ? uh[j] : 0.0;
        V[e] = Zq[e] + u;
        fz_a += L[e] * (Zq[e] * Zq[e]); fz_b += Qh[e] * Zq[e];
        fv_a += L[e] * (V[e] * V[e]);   fv_b += Qh[e] * V[e];
    }
    // element slots e whose eigenvalues are zero in every lane (low-rank constraints: all but the ends of the
    // spectrum): there 2 (1 + nu lam) == 2 exactly and the division in phi is a multiplication by 0.5
    unsigned nzmask = 0;
#pragma unroll
    for (int e = 0; e < EPL; e++)
        if (__builtin_amdgcn_ballot_w64(L[e] != 0.0) != 0ull) nzmask |= 1u << e;
    // violation of z itself (QuadraticFunction.violation, utilities.py:56-62) in eigen form
    const double fz = wave_sum(fz_a) + wave_sum(fz_b) + rk;
    if (lane == 0) {
        const double viol = (relop == RELOP_EQ) ? fabs(fz) : (fz > 0.0 ? fz : 0.0);
        atomicMax(&a.mvbits[r], (unsigned long long)__double_as_longlong(viol));   // viol >= 0: bit order = value order
    }
    if (a.viol_only) return;
    // onecons_qcqp(z + u, f): feasible inequality -> the point itself (utilities.py:157-158)
    const double fv = wave_sum(fv_a) + wave_sum(fv_b) + rk;
    double X[EPL];
    if (relop == RELOP_LE && fv <= 0.0) {
#pragma unroll
        for (int e = 0; e < EPL; e++) X[e] = V[e];
    } else {
        auto phi = [&](double nu) {
            double pa = 0.0, pb = 0.0;
#pragma unroll
            for (int e = 0; e < EPL; e++) {
                const double num = -(nu * Qh[e] - 2.0 * V[e]);
                const double xh = ((nzmask >> e) & 1u) ? admm_div(num, 2.0 * (1.0 + nu * L[e])) : num * 0.5;   // wave-uniform
                X[e] = xh;
                pa += L[e] * (xh * xh);
                pb += Qh[e] * xh;
            }
            return wave_sum(pa) + wave_sum(pb) + rk;
        };
        double s = a.slo[k], e_ = a.ehi[k];
        int guard = 0;
        if (s == -QM_INF) { s = -1.0; while (phi(s) <= 0.0 && guard++ < 2000) s *= 2.0; }
        if (e_ == QM_INF) { e_ = 1.0; while (phi(e_) >= 0.0 && guard++ < 4000) e_ *= 2.0; }
        int steps = 0;
        while (e_ - s > a.sec_tol && steps++ < 100000) {
            const double mid = (s + e_) / 2.0;
            const double p = phi(mid);
            if (p > 0.0) s = mid;
            else if (p < 0.0) e_ = mid;
            else { s = e_ = mid; break; }
        }
        (void)phi((s + e_) / 2.0);
    }
    // dual update in the eigenbasis and the operand of the consensus product
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        const int64_t j = lane + 64 * e;
        if (j < n) {
            const double u_old = a.first_iter ? 0.0 : uh[j];
            const double u_new = u_old + (Zq[e] - X[e]);   // us[i] += z - xs[i]
            uh[j] = u_new;
            zq[j] = X[e] - u_new;                           // x_k - u_k
        }
    }
}

// ---- small element-wise / per-restart kernels -------------------------------------------------

// z = S / m for active restarts (admm_phase1, qcqp.py:205)
__global__ void admm_z_phase1_kernel(double *Z, const double *S, const uint8_t *act, int64_t n, int64_t R,
                                     double md) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * R) return;
    if (act[idx / n]) Z[idx] = S[idx] / md;
}

// dst = alpha * src (element-wise); used for S = m x0 (xs = x0, us = 0)
__global__ void admm_scale_kernel(double *dst, const double *src, int64_t count, double alpha) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < count) dst[idx] = alpha * src[idx];
}

// dst[:, r] = pick[r] ? a[:, r] : b[:, r]
__global__ void admm_select_cols_kernel(double *dst, const double *a, const double *b, const uint8_t *pick,
                                        int64_t n, int64_t R) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * R) return;
    dst[idx] = pick[idx / n] ? a[idx] : b[idx];
}

// rhs = 2 rho S - q0 (qcqp.py:231)
__global__ void admm_rhs_kernel(double *RHS, const double *S, const double *q0, int64_t n, int64_t R, double rho) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * R) return;
    RHS[idx] = 2.0 * rho * S[idx] - q0[idx % n];
}

// Znew -> Z for active restarts, and ||Zlast - Znew||^2 per restart (one block per restart)
__global__ __launch_bounds__(256) void admm_take_z_kernel(double *Z, const double *Znew, const double *Zlast,
                                                           const uint8_t *act, double *dist2, int64_t n) {
    __shared__ double red[256];
    const int64_t r = blockIdx.x;
    double acc = 0.0;
    if (act[r]) {
        for (int64_t j = threadIdx.x; j < n; j += 256) {
            const double zn = Znew[r * n + j];
            const double d = Zlast[r * n + j] - zn;
            acc += d * d;
            Z[r * n + j] = zn;
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) dist2[r] = red[0];
}

// f0(z) = z.(Y + q0) + r0 with Y = P0 z (one block per restart)
__global__ __launch_bounds__(256) void admm_f0_kernel(const double *Z, const double *Y, const double *q0, double r0,
                                                       double *f0, int64_t n) {
    __shared__ double red[256];
    const int64_t r = blockIdx.x;
    double acc = 0.0;
    for (int64_t j = threadIdx.x; j < n; j += 256) acc += (Y[r * n + j] + q0[j]) * Z[r * n + j];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) f0[r] = red[0] + r0;
}

struct AdmmBook {
    int64_t n, R;
    int phase;                 // 1 or 2
    double tol, viol_lim;
    int have_last;             // phase 2: a previous z exists
    const double *Z;
    double *Zlast, *BEST;
    const double *dist2, *f0z;
    const unsigned long long *mvbits;
    double *best_f0, *best_mv;
    uint8_t *act;
    int64_t *iters;
    int *nactive;
};

// per-restart control flow of admm_phase1 (qcqp.py:202-204) / admm_phase2 (qcqp.py:240-249);
// one block per restart
__global__ __launch_bounds__(256) void admm_book_kernel(AdmmBook b) {
    const int64_t r = blockIdx.x;
    if (!b.act[r]) return;
    __shared__ int take;
    const double mv = __longlong_as_double((long long)b.mvbits[r]);
    if (threadIdx.x == 0) {
        take = 0;
        bool stop = false;
        if (b.phase == 1) {
            if (mv < b.tol) stop = true;                       // qcqp.py:203
        } else {
            if (b.have_last && sqrt(b.dist2[r]) < b.tol) stop = true;       // qcqp.py:241-242 (before bestx)
            else if (mv > b.viol_lim) stop = true;                          // qcqp.py:248
            else {
                // bestx = better(z, bestx) (utilities.py:135-146): first argument wins only if strictly better
                const long long v1 = (long long)(mv / 1e-4), v2 = (long long)(b.best_mv[r] / 1e-4);
                const double f1 = b.f0z[r], f2 = b.best_f0[r];
                if (v1 < v2 || (v1 == v2 && f1 < f2)) { take = 1; b.best_f0[r] = f1; b.best_mv[r] = mv; }
            }
        }
        if (stop) b.act[r] = 0;
        else { b.iters[r]++; atomicAdd(b.nactive, 1); }
    }
    __syncthreads();
    if (b.phase == 2) {
        for (int64_t j = threadIdx.x; j < b.n; j += 256) {
            const double z = b.Z[r * b.n + j];
            b.Zlast[r * b.n + j] = z;
            if (take) b.BEST[r * b.n + j] = z;
        }
    }
}

// S = sum_k C_k over the m batched products, fixed order
__global__ void admm_sum_k_kernel(const double *__restrict__ Cb, double *__restrict__ S, int m, int64_t tot) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= tot) return;
    double s = Cb[idx];
    for (int k = 1; k < m; k++) s += Cb[(int64_t)k * tot + idx];
    S[idx] = s;
}

}  // namespace qcqpmi
