// Consensus ADMM improve (qcqp.py:195-285) for a whole population, in the eigenbasis of each constraint.
//
// The reference keeps, per restart, one copy x_k and one dual u_k of the n-vector for each of the m constraints and
// per iteration calls onecons_qcqp(z + u_k, f_k) (utilities.py:149-196): rotate into the eigenbasis of P_k (Q_k^T v),
// solve the secular equation by bisection, rotate back (Q_k xhat).  Here the state lives IN the eigenbasis:
// uh_k = Q_k^T u_k.  Per iteration, for all restarts at once,
//     ZQ = [Q_1^T; ...; Q_m^T] Z                       gemm_pk_kernel   ((m n) x n by n x R)
//     per (k, r): vhat = ZQ_k + uh_k; xhat_k = secular(vhat);  uh_k += ZQ_k - xhat_k;  D_k = xhat_k - uh_k
//                                                      admm_secular_kernel
//     S = sum_k x_k - sum_k u_k = [Q_1 ... Q_m] D      gemm_pk_kernel   (n x (m n) by (m n) x R, K split over grid.z)
//     z = S / m  (phase 1)      z = (2 (P0 + rho m I))^-1 (2 rho S - q0)  (phase 2)      admm_zupdate_kernel (+ one gemm)
//     f0(z), violations, stop rules, `better`         admm_f0_kernel, admm_book_kernel
// which is algebraically the reference's iteration (Q_k orthogonal) without ever materialising x_k or u_k.  Every
// product runs on the engine's own fp64 MFMA GEMM (gemm_pk.h): no vendor BLAS on the iteration path.
//
// LOW-RANK constraints (beamforming: P_k = +-(a a' + b b'), rank 2).  onecons_qcqp moves v = z + u_k only inside
// span(B_k), B_k = [eigenvectors of the nonzero eigenvalues, the part of q_k outside them]: for lambda_j = 0 and
// qhat_j = 0 the eigenbasis formula gives xhat_j = vhat_j.  Hence u_k = B_k w_k stays in that span, w_k (rp numbers
// instead of n) is the whole dual state, and with vhat = B_k^T z + w_k
//     w_k <- vhat - xhat,        S = m z + sum_k B_k (2 xhat_k - vhat_k - B_k^T z)
// -- exactly the same iteration with (m rp) x n operators instead of (m n) x n: 0.34 GFLOP per restart-iteration
// at BASELINE configs[3] become 5 MFLOP.  The secular equation is then solved on the rp (<= 8) coordinates by one
// thread per (constraint, restart): admm_secular_small_kernel.
//
// Layout: everything tile-major like the population (kernels.h): Z [tile][n16][16]; the "hat" arrays ZQ / UH
// [tile][Mh16][16] with row h = k * rows_k + j (rows_k = n in the full basis, rp in a low-rank basis).  A wave owns one
// (constraint, restart) pair and keeps its share of the n coordinates in registers across the bisection; its loads
// are strided by 16 doubles, the 16 restarts of a tile run on neighbouring waves and share the cache lines.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "onevar.h"

namespace qcqpmi {

// threads of the per-tile streaming kernels (z-update, f0, book): one workgroup per tile of 16 restarts, 64 row groups --
// with 256 threads the 64 workgroups of 1024 restarts kept a quarter of the chip's memory pipes busy
constexpr int ADMM_TPB = 1024;

struct AdmmArgs {
    int64_t n, m, R;
    int64_t rows_k;       // hat rows per constraint (n or rp)
    int64_t Mh16;         // m * rows_k padded to a multiple of 16
    const double *lam;    // [m][rows_k] eigenvalues (numpy eigh order / basis order)
    const double *qhat;   // [m][rows_k] B_k^T q_k
    const double *rk;     // [m]
    const int *relop;     // [m]
    const double *slo;    // [m] bracket start  max_{lam>0} -1/lam  (or -inf)
    const double *ehi;    // [m] bracket end    min_{lam<0} -1/lam  (or +inf)
    double *ZQ;           // in: B_k^T z ; out: the operand of the consensus product
    double *UH;           // in/out: dual in the basis
    const uint8_t *act;   // [R] restart still iterating
    unsigned long long *mvbits;  // [R] max violation of z over the constraints, as ordered bits
    int first_iter;       // 1: xs = x0, us = 0: the duals are zero and UH is not read
    int viol_only;        // 1: only the violations of z are wanted (no update)
    int lowrank;          // 1: reduced basis (operand 2 xhat - vhat - zq, dual vhat - xhat)
    int project_only;     // 1: unit operator onecons_qcqp: out = xhat (no dual, no update)
    double sec_tol;       // 1e-6 (utilities.py:149)
    int zq_planes;        // ZQ arrives as this many split-K partial planes (small kernel only; summed in a fixed order)
    int64_t zq_plane;     // doubles between two planes
    int no_shortcut;      // 1: every step of the bisection evaluates the secular function (cross-check of the one-row shortcut)
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

// tile-major address of hat row h of restart r
__device__ inline int64_t admm_hat_index(const AdmmArgs &a, int64_t r, int64_t h) {
    return ((r >> 4) * a.Mh16 + h) * 16 + (r & 15);
}

// NW = 1: one wave per (constraint k, restart r), four pairs per workgroup; EPL = elements per lane (rows_k <= 64 EPL).
// NW = 4 (rows_k > 4096, round 3): the whole workgroup of four waves shares one pair -- element j of thread t is t + 256 e,
// the sums of the secular function go through LDS (two buffers in turn: one barrier per sum), every thread sees the same
// totals in the same order, so the bisection takes the same branches in all four waves.
template <int EPL, int NW = 1>
__global__ __launch_bounds__(256) void admm_secular_kernel(AdmmArgs a) {
    __shared__ double sec_red[2][2][4];
    int sec_buf = 0;
    const int lane = (NW == 1) ? (threadIdx.x & 63) : (int)threadIdx.x;      // element slot of this thread
    constexpr int STRIDE = 64 * NW;
    const int64_t widx = (NW == 1) ? (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6) : (int64_t)blockIdx.x;
    if (widx >= a.m * ((a.R + 15) / 16) * 16) return;
    // sum over the pair's threads of (x, y): one wave (DPP) or the workgroup (DPP per wave, then LDS in wave order)
    auto pair_sum = [&](double x, double y, double *sx, double *sy) {
        const double wx = wave_sum(x), wy = wave_sum(y);
        if (NW == 1) { *sx = wx; *sy = wy; return; }
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sec_red[sec_buf][0][w] = wx; sec_red[sec_buf][1][w] = wy; }
        __syncthreads();
        double tx = sec_red[sec_buf][0][0], ty = sec_red[sec_buf][1][0];
#pragma unroll
        for (int q = 1; q < 4; q++) { tx += sec_red[sec_buf][0][q]; ty += sec_red[sec_buf][1][q]; }
        sec_buf ^= 1;
        *sx = tx; *sy = ty;
    };
    // consecutive waves: the 16 restarts of a tile for one constraint (they share every cache line they touch)
    const int64_t tile = widx / (16 * a.m), rem = widx % (16 * a.m);
    const int64_t k = rem / 16, r = tile * 16 + (rem % 16);
    if (r >= a.R || !a.act[r]) return;
    const int64_t n = a.rows_k;
    const int64_t base = admm_hat_index(a, r, k * n);
    double *zq = a.ZQ + base;
    double *uh = a.UH + base;
    const double *lm = a.lam + k * n, *qh = a.qhat + k * n;
    const double rk = a.rk[k];
    const int relop = a.relop[k];

    double L[EPL], Qh[EPL], V[EPL], Zq[EPL];
    double fz_a = 0.0, fz_b = 0.0, fv_a = 0.0, fv_b = 0.0;
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        const int64_t j = lane + STRIDE * e;
        const bool ok = j < n;
        L[e] = ok ? lm[j] : 0.0;
        Qh[e] = ok ? qh[j] : 0.0;
        Zq[e] = ok ? zq[j * 16] : 0.0;
        const double u = (ok && !a.first_iter && !a.project_only) ? uh[j * 16] : 0.0;
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
    double sa_, sb_;
    pair_sum(fz_a, fz_b, &sa_, &sb_);
    const double fz = sa_ + sb_ + rk;
    if (lane == 0 && a.mvbits) {
        const double viol = (relop == RELOP_EQ) ? fabs(fz) : (fz > 0.0 ? fz : 0.0);
        atomicMax(&a.mvbits[r], (unsigned long long)__double_as_longlong(viol));   // viol >= 0: bit order = value order
    }
    if (a.viol_only) return;
    // onecons_qcqp(z + u, f): feasible inequality -> the point itself (utilities.py:157-158)
    pair_sum(fv_a, fv_b, &sa_, &sb_);
    const double fv = sa_ + sb_ + rk;
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
            double ta, tb;
            pair_sum(pa, pb, &ta, &tb);
            return ta + tb + rk;
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
    // dual update in the basis and the operand of the consensus product
#pragma unroll
    for (int e = 0; e < EPL; e++) {
        const int64_t j = lane + STRIDE * e;
        if (j < n) {
            if (a.project_only) { zq[j * 16] = X[e]; continue; }
            const double u_old = a.first_iter ? 0.0 : uh[j * 16];
            const double u_new = u_old + (Zq[e] - X[e]);   // us[i] += z - xs[i]
            uh[j * 16] = u_new;
            zq[j * 16] = a.lowrank ? (2.0 * X[e] - V[e] - Zq[e]) : (X[e] - u_new);   // x_k - u_k (minus z in a reduced basis)
        }
    }
}

// The projection itself (utilities.py:165-196) on the RP coordinates of a reduced basis for one (constraint, restart) pair: v -> x.
// Shared by admm_secular_small_kernel and admm_unit_step_kernel: same expressions, same bits.
template <int RP>
__device__ __attribute__((always_inline)) inline void admm_small_solve(const AdmmArgs &a, int64_t k, const double (&L)[RP], const double (&Qh)[RP],
                                                                       const double (&V)[RP], double rk, int relop, double fv, double (&X)[RP]) {
    if (relop == RELOP_LE && fv <= 0.0) {
#pragma unroll
        for (int e = 0; e < RP; e++) X[e] = V[e];
    } else {
        auto phi = [&](double nu) {
            double p = 0.0;
#pragma unroll
            for (int e = 0; e < RP; e++) {
                const double num = -(nu * Qh[e] - 2.0 * V[e]);
                const double xh = (L[e] != 0.0) ? admm_div(num, 2.0 * (1.0 + nu * L[e])) : num * 0.5;
                X[e] = xh;
                p += L[e] * (xh * xh) + Qh[e] * xh;
            }
            return p + rk;
        };
        // One row (RP == 1: a constraint on ONE coordinate): the secular function is scalar,
        //   phi(nu) = L x^2 + q x + r,  x = (2 v - nu q) / (2 (1 + nu L)),  phi' = -(2 L x + q)^2 / (2 (1 + nu L)) <= 0 on the bracket,
        // so its root nu* is known in closed form (x* = the root of the constraint's own quadratic on v's side, nu* = 2 (v - x*) /
        // (2 L x* + q)) and every comparison the reference's loops make (utilities.py:176-194: the doubling of the bracket, the
        // bisection to 1e-6) is decided by the SIDE of nu* the trial value lies on -- the same outcome, without evaluating phi --
        // unless the trial is so close to nu* that rounding could decide (then phi is evaluated like before).  ~25-60 evaluations with
        // a division each become ~1 per solve: this kernel was 38 % of an ADMM iteration on Boolean least squares at n = 1024.
        bool fast = false;
        double nus = 0.0, dphi = 0.0, scale = 0.0;
        if (RP == 1 && !a.no_shortcut) {
            const double l = L[0], qh = Qh[0], v = V[0];
            const double w = qh + 2.0 * l * v;                 // sign of (1 + nu L) (2 L x + q): picks the root on the bracket's branch
            double xs = 0.0;
            bool ok = false;
            if (l != 0.0) {
                const double disc = qh * qh - 4.0 * l * rk;
                if (disc > 0.0 && w != 0.0) {
                    const double sq = sqrt(disc);
                    xs = (w > 0.0) ? (-qh + sq) / (2.0 * l) : (-qh - sq) / (2.0 * l);
                    ok = true;
                }
            } else if (qh != 0.0) { xs = -rk / qh; ok = true; }
            if (ok) {
                const double den = 2.0 * l * xs + qh;
                nus = 2.0 * (v - xs) / den;
                const double onel = 1.0 + nus * l;
                dphi = -(den * den) / (2.0 * onel);
                scale = fabs(l) * xs * xs + fabs(qh * xs) + fabs(rk);
                fast = den != 0.0 && onel > 0.0 && nus - nus == 0.0 && dphi - dphi == 0.0;
            }
        }
        // The sign of phi at a trial value: by the SIDE of nu* unless the trial is too close to tell (then phi is evaluated).  Written
        // straight-line -- `up` / `dn` are plain compares, phi sits behind ONE wave-uniform test (no lane of the wave near nu*: the common
        // case has nothing to skip over), the bracket moves by selects: with an early return per case a step was ten exec-mask regions
        // and 40 instructions, and the ~25 steps of a solve were what admm_unit_step_kernel spent its time ISSUING (a wave64 vector
        // instruction occupies its SIMD for 4 cycles whatever it does; round 6, profiles/r06_admm_unit_bases.md).  Same decisions, same
        // bits: |d| > c1 and |phi'(nu*) d| > c2  <=>  |d| > zone up to the rounding of c2 / |phi'| -- a trial THAT close to the zone's
        // edge has the sign of its side either way (the zone is 1e6 roundings wide).
        const double zone = fast ? fmax(1e-8 * (1.0 + fabs(nus)), 1e-10 * scale / fabs(dphi)) : QM_INF;
        bool pnan = false;                                     // phi was evaluated and is not a number (ends the doubling loops like the reference's comparisons do)
        auto psign = [&](double nu, bool &up, bool &dn) {
            const double d = nu - nus;
            const bool near = !(fabs(d) > zone);               // (also for a not-a-number trial)
            up = d < 0.0; dn = !up;
            if (__builtin_amdgcn_ballot_w64(near) != 0ull) {
                if (near) { const double p = phi(nu); up = p > 0.0; dn = p < 0.0; pnan = p != p; }
            }
        };
        double s = a.slo[k], e_ = a.ehi[k];
        int guard = 0;
        bool up, dn;
        if (s == -QM_INF) { s = -1.0; for (;;) { psign(s, up, dn); if (up || pnan || guard++ >= 2000) break; s *= 2.0; } }          // while phi(s) <= 0
        if (e_ == QM_INF) { e_ = 1.0; for (;;) { psign(e_, up, dn); if (dn || pnan || guard++ >= 4000) break; e_ *= 2.0; } }        // while phi(e) >= 0
        int steps = 0;
        while (e_ - s > a.sec_tol && steps++ < 100000) {
            const double mid = (s + e_) / 2.0;
            psign(mid, up, dn);
            // phi > 0: s = mid; phi < 0: e = mid; otherwise (zero, or not a number) both, which ends the loop (utilities.py:189-194)
            s = (up || !dn) ? mid : s;
            e_ = (dn || !up) ? mid : e_;
        }
        (void)phi((s + e_) / 2.0);
    }
}

// The same for a reduced basis of RP <= 8 coordinates: one THREAD per (constraint, restart).
template <int RP>
__global__ __launch_bounds__(256) void admm_secular_small_kernel(AdmmArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // consecutive threads: the 16 restarts of a tile, then the next constraint (coalesced 128-byte rows)
    const int64_t tile = idx / (16 * a.m), rem = idx % (16 * a.m);
    const int64_t k = rem / 16, r = tile * 16 + (rem % 16);
    const bool valid = idx < ((a.R + 15) / 16) * 16 * a.m && r < a.R && a.act[r];
    // max violation per restart: with m a multiple of 16 a workgroup is 16 constraints x the 16 restarts of ONE tile -- the maximum
    // is taken in LDS first and one thread per restart goes to memory (a global atomic per (constraint, restart) pair -- 4 M per
    // launch at n = m = 1024, 4096 restarts -- WAS the kernel's time: 194 us of which the bisections are a fraction)
    __shared__ unsigned long long mvs[16];
    const bool blockred = a.mvbits != nullptr && (a.m % 16) == 0;
    if (blockred) {
        if (threadIdx.x < 16) mvs[threadIdx.x] = 0ull;
        __syncthreads();
    }
    const int64_t base = valid ? admm_hat_index(a, r, k * RP) : 0;
    double *zq = a.ZQ + base;
    double *uh = a.UH + base;
    const double *lm = a.lam + (valid ? k * RP : 0), *qh = a.qhat + (valid ? k * RP : 0);
    const double rk = valid ? a.rk[k] : 0.0;
    const int relop = valid ? a.relop[k] : RELOP_LE;
    double L[RP], Qh[RP], V[RP], Zq[RP], X[RP];
    double fz = 0.0, fv = 0.0;
    if (valid) {
#pragma unroll
        for (int e = 0; e < RP; e++) {
            L[e] = lm[e]; Qh[e] = qh[e];
            double zsum = zq[e * 16];
            for (int z = 1; z < a.zq_planes; z++) zsum += zq[e * 16 + z * a.zq_plane];
            Zq[e] = zsum;
            const double u = (!a.first_iter && !a.project_only) ? uh[e * 16] : 0.0;
            V[e] = Zq[e] + u;
            fz += L[e] * (Zq[e] * Zq[e]) + Qh[e] * Zq[e];
            fv += L[e] * (V[e] * V[e]) + Qh[e] * V[e];
        }
        fz += rk; fv += rk;
        if (a.mvbits) {
            const double viol = (relop == RELOP_EQ) ? fabs(fz) : (fz > 0.0 ? fz : 0.0);
            if (blockred) atomicMax(&mvs[threadIdx.x & 15], (unsigned long long)__double_as_longlong(viol));
            else atomicMax(&a.mvbits[r], (unsigned long long)__double_as_longlong(viol));
        }
    }
    if (blockred) {
        __syncthreads();
        if (threadIdx.x < 16 && valid && mvs[threadIdx.x] != 0ull) atomicMax(&a.mvbits[r], mvs[threadIdx.x]);
    }
    if (!valid || a.viol_only) return;
    admm_small_solve<RP>(a, k, L, Qh, V, rk, relop, fv, X);
#pragma unroll
    for (int e = 0; e < RP; e++) {
        if (a.project_only) { zq[e * 16] = X[e]; continue; }
        uh[e * 16] = V[e] - X[e];
        zq[e * 16] = 2.0 * X[e] - V[e] - Zq[e];
    }
}

// Unit bases with one row per constraint (separable constraints: p x_i^2 + q x_i + r ~ 0), round 5: gather, projection and
// scatter of one ADMM iteration in ONE pass -- a thread per (coordinate i, restart): for every constraint h on the coordinate
//     zq = s_h Z[i]  (admm_unit_gather_kernel),  the pair (h, restart) of admm_secular_small_kernel<1>,  S[i] += s_h d_h  (admm_unit_scatter_kernel)
// with the same expressions in the same order, so the three-kernel path (debug bit 2) gives the same bits.  Z and the duals are read
// once, the operand rows never go to memory: 140 us of launches per iteration at n = m = 1024, 4096 restarts become one.
// zmode 1 (phase 1): the z-update of the iteration in front -- z = (S + m z) / m (admm_zupdate_kernel, qcqp.py:205) from the sum this
// thread wrote in the previous iteration, written to Z before it is used; zmode 2 (phase 2, dense solve): the right-hand side of the NEXT
// iteration's solve behind -- Y = 2 rho (S + m z) - q0 (qcqp.py:231) -- instead of S; zmode 0: S only.  Either way the z-update launch
// and its pass over Z and S are gone; the values are those of admm_zupdate_kernel (same expressions).
__global__ __launch_bounds__(256) void admm_unit_step_kernel(AdmmArgs a, double *Z, double *S, const int *__restrict__ uptr,
                                                             const int *__restrict__ ulist, const double *__restrict__ usgn, int64_t n16, int64_t total,
                                                             int zmode, int add_mz, double mm, double rho, const double *__restrict__ q0, double *Y, int64_t n) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // element (tile, coordinate, column): a workgroup = 16 coordinates of ONE tile
    __shared__ unsigned long long mvs[16];
    if (threadIdx.x < 16) mvs[threadIdx.x] = 0ull;
    __syncthreads();
    const int64_t col = e & 15, i = (e >> 4) % n16, t = (e >> 4) / n16, r = t * 16 + col;
    const bool valid = e < total && r < a.R && a.act[r];
    double acc = 0.0;
    if (valid) {
        double z = Z[e];
        if (zmode == 1 && i < n) {
            double sm = S[e];
            if (add_mz) sm += mm * z;
            z = sm / mm;                                 // qcqp.py:205
            Z[e] = z;
        }
        for (int q = uptr[i]; q < uptr[i + 1]; q++) {
            const int64_t h = ulist[q];                  // hat row = constraint (one row per constraint)
            const double sg = usgn[h];
            double *uh = a.UH + admm_hat_index(a, r, h);
            double L[1], Qh[1], V[1], Zq[1], X[1];
            L[0] = a.lam[h]; Qh[0] = a.qhat[h];
            const double rk = a.rk[h];
            const int relop = a.relop[h];
            Zq[0] = sg * z;
            const double u = a.first_iter ? 0.0 : uh[0];
            V[0] = Zq[0] + u;
            double fz = 0.0, fv = 0.0;
            fz += L[0] * (Zq[0] * Zq[0]) + Qh[0] * Zq[0];
            fv += L[0] * (V[0] * V[0]) + Qh[0] * V[0];
            fz += rk; fv += rk;
            const double viol = (relop == RELOP_EQ) ? fabs(fz) : (fz > 0.0 ? fz : 0.0);
            atomicMax(&mvs[threadIdx.x & 15], (unsigned long long)__double_as_longlong(viol));
            admm_small_solve<1>(a, h, L, Qh, V, rk, relop, fv, X);
            uh[0] = V[0] - X[0];
            acc += sg * (2.0 * X[0] - V[0] - Zq[0]);
        }
        if (zmode == 2 && i < n) {
            double sm = acc;
            if (add_mz) sm += mm * z;
            Y[e] = 2.0 * rho * sm - q0[i];               // qcqp.py:231
        }
    }
    if (e < total && zmode != 2) S[e] = acc;
    __syncthreads();
    if (threadIdx.x < 16 && valid && mvs[threadIdx.x] != 0ull) atomicMax(&a.mvbits[r], mvs[threadIdx.x]);
}

// ---- per-tile kernels on the tile-major n x R arrays: 256 threads = 16 restarts x 16 row lanes ----------------------

// dst = alpha * src (whole padded array)
__global__ void admm_scale_kernel(double *dst, const double *src, int64_t count, double alpha) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < count) dst[idx] = alpha * src[idx];
}

// dst[:, r] = pick[r] ? a[:, r] : b[:, r]
__global__ void admm_select_cols_kernel(double *dst, const double *a, const double *b, const uint8_t *pick,
                                        int64_t n16, int64_t R, int64_t count) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const int64_t r = (idx / (n16 * 16)) * 16 + (idx & 15);
    dst[idx] = (r < R && pick[r]) ? a[idx] : b[idx];
}

struct AdmmZArgs {
    int64_t n, n16, R;
    int phase;            // 1: z = S / m ; 2: rhs = 2 rho S - q0, then z = dinv * rhs (diagonal) or rhs -> Y (dense solve follows)
    int zs;               // partial planes of S = W D to be summed (fixed order)
    int64_t plane;        // doubles per plane
    int add_mz;           // reduced basis: S = m z + W D
    double m, rho;
    const double *Sp;     // [zs][tile][n16][16]
    double *Z;            // current z (updated in place for active restarts when the solve is diagonal / phase 1)
    double *Y;            // phase 2, dense solve: rhs
    const double *q0;     // [n16]
    const double *dinv;   // [n16] 1 / (2 (P0_ii + rho m)) for a diagonal P0, or nullptr
    const double *Zlast;  // phase 2
    double *dist2;        // [R] ||Zlast - Znew||^2 (phase 2, diagonal solve)
    const uint8_t *act;
    // phase 2, diagonal solve: f0 of the new z in the same pass (same expression and summation order as admm_f0_kernel)
    const double *pdiag;  // [n16] P0_ii, or nullptr: no f0 here
    double r0;
    double *f0z;          // [R]
};

// one workgroup per tile: sums the planes of the consensus product and performs the z-update that needs no solve
__global__ __launch_bounds__(ADMM_TPB) void admm_zupdate_kernel(AdmmZArgs a) {
    __shared__ double red[ADMM_TPB];
    const int64_t tile = blockIdx.x;
    const int rr = threadIdx.x & 15, jl = threadIdx.x >> 4;
    const int64_t r = tile * 16 + rr;
    const bool on = r < a.R && a.act[r];
    double acc = 0.0, accf = 0.0;
    for (int64_t j = jl; j < a.n16; j += ADMM_TPB / 16) {
        const int64_t idx = (tile * a.n16 + j) * 16 + rr;
        double s = a.Sp[idx];
        for (int z = 1; z < a.zs; z++) s += a.Sp[(int64_t)z * a.plane + idx];
        if (a.add_mz) s += a.m * a.Z[idx];
        if (!on || j >= a.n) continue;
        if (a.phase == 1) {
            a.Z[idx] = s / a.m;                              // qcqp.py:205
        } else {
            const double rhs = 2.0 * a.rho * s - a.q0[j];    // qcqp.py:231
            if (a.dinv) {
                const double zn = a.dinv[j] * rhs;
                const double d = a.Zlast[idx] - zn;
                acc += d * d;
                a.Z[idx] = zn;
                if (a.pdiag) accf += (a.pdiag[j] * zn + a.q0[j]) * zn;
            } else {
                a.Y[idx] = rhs;
            }
        }
    }
    if (a.phase == 2 && a.dinv) {
        red[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < 16) {
            double s = 0.0;
            for (int g = 0; g < ADMM_TPB / 16; g++) s += red[g * 16 + threadIdx.x];
            if (tile * 16 + threadIdx.x < a.R) a.dist2[tile * 16 + threadIdx.x] = s;
        }
        if (a.pdiag) {
            __syncthreads();
            red[threadIdx.x] = accf;
            __syncthreads();
            if (threadIdx.x < 16) {
                double s = 0.0;
                for (int g = 0; g < ADMM_TPB / 16; g++) s += red[g * 16 + threadIdx.x];
                const int64_t rq = tile * 16 + threadIdx.x;
                if (rq < a.R && a.act[rq]) a.f0z[rq] = s + a.r0;
            }
        }
    }
}

// dense solve: Znew = Minv rhs came out of the GEMM; Z <- Znew for active restarts, ||Zlast - Znew||^2 per restart.
// With rhs != nullptr also f0(z) for `better` (qcqp.py:249) WITHOUT a second n x n product: z solves 2 (P0 + rho m I) z = rhs
// (qcqp.py:231-232), hence P0 z = rhs / 2 - rho m z and f0(z) = z.(rhs / 2 - rho m z + q0) + r0 -- exact up to the accuracy of
// the solve (Newton-Schulz residual < 1e-10; the points, their reported objective and the final `better` are evaluated with
// P0 itself).
__global__ __launch_bounds__(ADMM_TPB) void admm_take_z_kernel(double *Z, const double *Znew, const double *Zlast,
                                                           const uint8_t *act, double *dist2, int64_t n, int64_t n16, int64_t R,
                                                           const double *rhs, const double *q0, double r0, double rho_m, double *f0) {
    __shared__ double red[ADMM_TPB], redf[ADMM_TPB];
    const int64_t tile = blockIdx.x;
    const int rr = threadIdx.x & 15, jl = threadIdx.x >> 4;
    const int64_t r = tile * 16 + rr;
    const bool on = r < R && act[r];
    double acc = 0.0, accf = 0.0;
    if (on)
        for (int64_t j = jl; j < n; j += ADMM_TPB / 16) {
            const int64_t idx = (tile * n16 + j) * 16 + rr;
            const double zn = Znew[idx];
            const double d = Zlast[idx] - zn;
            acc += d * d;
            Z[idx] = zn;
            if (rhs) accf += ((0.5 * rhs[idx] - rho_m * zn) + q0[j]) * zn;
        }
    red[threadIdx.x] = acc;
    redf[threadIdx.x] = accf;
    __syncthreads();
    if (threadIdx.x < 16) {
        double s = 0.0, sf = 0.0;
        for (int g = 0; g < ADMM_TPB / 16; g++) { s += red[g * 16 + threadIdx.x]; sf += redf[g * 16 + threadIdx.x]; }
        if (tile * 16 + threadIdx.x < R) {
            dist2[tile * 16 + threadIdx.x] = s;
            if (rhs && act[tile * 16 + threadIdx.x]) f0[tile * 16 + threadIdx.x] = sf + r0;
        }
    }
}

// f0(z) = z.(Y + q0) + r0 with Y = P0 z from the GEMM, or Y_j = d_j z_j for a diagonal P0 (pdiag != nullptr)
__global__ __launch_bounds__(ADMM_TPB) void admm_f0_kernel(const double *Z, const double *Y, const double *pdiag, const double *q0,
                                                       double r0, double *f0, int64_t n, int64_t n16, int64_t R) {
    __shared__ double red[ADMM_TPB];
    const int64_t tile = blockIdx.x;
    const int rr = threadIdx.x & 15, jl = threadIdx.x >> 4;
    double acc = 0.0;
    for (int64_t j = jl; j < n; j += ADMM_TPB / 16) {
        const int64_t idx = (tile * n16 + j) * 16 + rr;
        const double z = Z[idx];
        const double y = pdiag ? pdiag[j] * z : Y[idx];
        acc += (y + q0[j]) * z;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        double s = 0.0;
        for (int g = 0; g < ADMM_TPB / 16; g++) s += red[g * 16 + threadIdx.x];
        if (tile * 16 + threadIdx.x < R) f0[tile * 16 + threadIdx.x] = s + r0;
    }
}

struct AdmmBook {
    int64_t n, n16, R;
    int phase;                 // 1 or 2
    double tol, viol_lim;
    int have_last;             // phase 2: a previous z exists
    const double *Z;
    double *Zlast, *BEST;
    const double *dist2, *f0z;
    unsigned long long *mvbits;
    double *best_f0, *best_mv;
    uint8_t *act;
    int64_t *iters;
    int *nactive;
};

// per-restart control flow of admm_phase1 (qcqp.py:202-204) / admm_phase2 (qcqp.py:240-249); one workgroup per tile
__global__ __launch_bounds__(ADMM_TPB) void admm_book_kernel(AdmmBook b) {
    __shared__ int take[16], live[16];
    const int64_t tile = blockIdx.x;
    const int rr = threadIdx.x & 15, jl = threadIdx.x >> 4;
    if (threadIdx.x < 16) {
        const int64_t r = tile * 16 + threadIdx.x;
        int tk = 0, lv = 0;
        if (r < b.R && b.act[r]) {
            const double mv = __longlong_as_double((long long)b.mvbits[r]);
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
                    if (v1 < v2 || (v1 == v2 && f1 < f2)) { tk = 1; b.best_f0[r] = f1; b.best_mv[r] = mv; }
                }
            }
            if (stop) b.act[r] = 0;
            else { b.iters[r]++; atomicAdd(b.nactive, 1); }
            lv = 1;
        }
        if (r < b.R) b.mvbits[r] = 0ull;      // read: the next iteration's projections start from zero (no fill launch per iteration)
        take[threadIdx.x] = tk; live[threadIdx.x] = lv;
    }
    __syncthreads();
    if (b.phase == 2 && live[rr]) {
        const bool tk = take[rr] != 0;
        for (int64_t j = jl; j < b.n; j += ADMM_TPB / 16) {
            const int64_t idx = (tile * b.n16 + j) * 16 + rr;
            const double z = b.Z[idx];
            b.Zlast[idx] = z;
            if (tk) b.BEST[idx] = z;
        }
    }
}

// onecons unit operator: keep the hat rows of constraint k only.  Full basis: zero the others (x = Q_k xhat).  Reduced
// basis: rows of k become xhat - B_k^T z, the others zero (x = z + B_k (xhat - B_k^T z)).
__global__ void admm_onecons_mask_kernel(double *ZQ, const double *ZB, int64_t Mh16, int64_t rows, int64_t k, int64_t count, int lowrank) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    const int64_t h = (idx / 16) % Mh16;
    const bool mine = h >= k * rows && h < (k + 1) * rows;
    ZQ[idx] = mine ? (lowrank ? ZQ[idx] - ZB[idx] : ZQ[idx]) : 0.0;
}

// y += x
__global__ void admm_axpy_kernel(double *y, const double *x, int64_t count) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < count) y[idx] += x[idx];
}

// out[k][i][c] = sum_j P_k[i][j] Vin[(shared ? 0 : k)][j][c]   (p columns, row-major n x p blocks): the two passes over the
// constraint matrices of the rank-revealing setup (range probe Y = P Omega, then Z = P Q).  One workgroup per
// (constraint, 16 rows): thread (row, column).
__global__ __launch_bounds__(256) void admm_apply_constraints_kernel(const double *__restrict__ gP, const double *__restrict__ Vin,
                                                                      double *__restrict__ out, int64_t n, int p, int shared) {
    const int64_t k = blockIdx.y, i0 = (int64_t)blockIdx.x * 16;
    const int c = threadIdx.x & 15, il = threadIdx.x >> 4;
    const int64_t i = i0 + il;
    if (i >= n) return;
    const double *Pr = gP + (k * n + i) * n;
    const double *V = Vin + (shared ? 0 : k * n * p);
    for (int c0 = 0; c0 < p; c0 += 16) {
        if (c0 + c >= p) continue;
        double s0 = 0.0, s1 = 0.0;
        int64_t j = 0;
        for (; j + 1 < n; j += 2) {
            s0 = __builtin_fma(Pr[j], V[j * p + c0 + c], s0);
            s1 = __builtin_fma(Pr[j + 1], V[(j + 1) * p + c0 + c], s1);
        }
        if (j < n) s0 = __builtin_fma(Pr[j], V[j * p + c0 + c], s0);
        out[(k * n + i) * p + c0 + c] = s0 + s1;
    }
}

}  // namespace qcqpmi
