// admm_fused_kernel -- the whole improve_admm (qcqp.py:254-285) for a tile of 16 restarts inside one persistent kernel.
//
//   reference                                    here
//   admm_phase1  qcqp.py:195-212                  phase loop with PHASE = 1
//   admm_phase2  qcqp.py:215-251                  phase loop with PHASE = 2 (bestx bookkeeping :241-249)
//   onecons_qcqp utilities.py:149-196             secular_pair<RP> (same bracket / bisection to 1e-6 as admm.h)
//   better       utilities.py:135-146             better_first, between the phases and at the end (qcqp.py:281, 284)
//
// Formulation: the reduced bases of admm.h (low-rank constraints: the dual of constraint k lives in span(B_k), rp <= 8
// numbers) with a diagonal P0 -- BASELINE.json configs[3] (beamforming: P0 = I, rank-2 constraints).  Other problems keep
// the multi-launch path of capi_admm.inc, which this kernel reproduces to rounding (same expressions, other summation
// order in the two products) and which stays available as the cross-check (qcqpmi_admm_fused(ctx, 0)).
//
// Work split.  A tile of 16 restarts is owned by a CLUSTER of C workgroups (C = 1, 2, 4, 8, 16; chosen by the host so
// that tiles x C fills the chip: 1024 restarts = 64 tiles would otherwise leave 192 of 256 CUs idle).  Member c owns
//   * the rows [16 b_lo, 16 b_hi) of z (blocks of 16 split C ways) for all 16 restarts, resident in LDS for the whole run,
//   * the constraints [k_lo, k_hi) (split C ways): their duals uh (LDS) and their secular solves.
// One iteration =
//   1. z-update of the own rows: T = W[rows, :] d on the matrix cores (v_mfma_f64_16x16x4_f64, A fragments streamed from
//      L2, B = the operand rows d in LDS), then element-wise in the accumulator registers
//           phase 1: z = (m z + T) / m        phase 2: z = (2 rho (m z + T) - q0) / (2 (P0_ii + rho m))
//      with ||z_old - z||^2 and f0(z) accumulated on the fly (partial over the own rows);
//   2. partial ZQ = W[rows, :]^T z[rows] on the matrix cores (B = the z slice in LDS), written to the member's slot of
//      the cluster's exchange buffer together with the two partial sums;                                  [exchange 1]
//   3. every member sums the C partials of ITS constraints' rows in a fixed order, solves the secular equation of every
//      (own constraint, restart) pair -- one thread per pair --, updates the duals, publishes the operand rows
//      d = 2 xhat - vhat - zq and its partial max violation;                                             [exchange 2]
//   4. every member reads all operand rows into LDS and runs the per-restart control flow of the reference (stop rules,
//      bestx = better(z, bestx)) redundantly -- identical inputs in identical order, hence identical decisions.
// The exchanges are agent-scope atomics on global memory (data: relaxed stores / loads, flags: release / acquire of a
// sequence number per member), so the members of a cluster may sit on different XCDs; the block index is arranged so that
// they normally share one (blocks are dealt round-robin to the 8 XCDs) and the traffic stays in one L2.  The launch is
// cooperative (co-residency guaranteed), spin waits are bounded and raise an abort flag instead of hanging.
#include "admm_fused.h"

#include "onevar.h"

namespace qcqpmi {
namespace {

typedef double af_v4d __attribute__((ext_vector_type(4)));

__device__ inline double ag_load(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void ag_store(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// num / den through the hardware reciprocal + two Newton steps: the expression of admm.h (admm_div), so that the fused
// and the multi-launch paths bisect on bit-identical secular functions
__device__ inline double af_div(double num, double den) {
    double r = __builtin_amdgcn_rcp(den);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    return num * r;
}

// QCQPForm.better (utilities.py:135-146): does it return its FIRST argument?  (strictly better only)
__device__ inline bool better_first(double f1, double v1, double f2, double v2) {
    const long long b1 = (long long)(v1 / 1e-4), b2 = (long long)(v2 / 1e-4);
    if (b1 < b2) return true;
    if (b2 < b1) return false;
    return f1 < f2;
}

struct AfState {                     // per-tile scalars in LDS
    double dist2[16], f0z[16], mvv[16], best_f0[16], best_mv[16];
    double fx0[16], vx0[16], fx1[16], vx1[16], pd2[16], pf0[16];
    unsigned long long mvbits[16];
    long long it1[16], it2[16];
    int act[16], take[16], live[16];
    int nactive, abort;
    double red[8][2][16];
    double scr[512];
};

// onecons_qcqp on the rp coordinates of a reduced basis for ONE (constraint, restart) pair: admm_secular_small_kernel of
// admm.h, with the operands in LDS.  zq / uh: column of the pair, rows 16 doubles apart; dout: the same in global memory.
template <int RP>
__device__ inline void secular_pair(const AdmmFusedArgs &a, int k, const double *zq, double *uh, int first_iter, int viol_only,
                                    unsigned long long *mvslot, double *dout) {
    const double *lm = a.lam + (int64_t)k * RP, *qh = a.qhat + (int64_t)k * RP;
    const double rk = a.rk[k];
    const int relop = a.relop[k];
    double L[RP], Qh[RP], V[RP], Zq[RP], X[RP];
    double fz = 0.0, fv = 0.0;
#pragma unroll
    for (int e = 0; e < RP; e++) {
        L[e] = lm[e]; Qh[e] = qh[e];
        Zq[e] = zq[e * 16];
        const double u = (!first_iter && !viol_only) ? uh[e * 16] : 0.0;
        V[e] = Zq[e] + u;
        fz += L[e] * (Zq[e] * Zq[e]) + Qh[e] * Zq[e];
        fv += L[e] * (V[e] * V[e]) + Qh[e] * V[e];
    }
    fz += rk; fv += rk;
    {
        const double viol = (relop == RELOP_EQ) ? fabs(fz) : (fz > 0.0 ? fz : 0.0);
        atomicMax(mvslot, (unsigned long long)__double_as_longlong(viol));     // viol >= 0: bit order = value order
    }
    if (viol_only) return;
    if (relop == RELOP_LE && fv <= 0.0) {
#pragma unroll
        for (int e = 0; e < RP; e++) X[e] = V[e];
    } else {
        auto phi = [&](double nu) {
            double p = 0.0;
#pragma unroll
            for (int e = 0; e < RP; e++) {
                const double num = -(nu * Qh[e] - 2.0 * V[e]);
                const double xh = (L[e] != 0.0) ? af_div(num, 2.0 * (1.0 + nu * L[e])) : num * 0.5;
                X[e] = xh;
                p += L[e] * (xh * xh) + Qh[e] * xh;
            }
            return p + rk;
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
#pragma unroll
    for (int e = 0; e < RP; e++) {
        uh[e * 16] = V[e] - X[e];
        ag_store(dout + e * 16, 2.0 * X[e] - V[e] - Zq[e]);
    }
}

__device__ inline void secular_dispatch(const AdmmFusedArgs &a, int k, const double *zq, double *uh, int first_iter,
                                        int viol_only, unsigned long long *mvslot, double *dout) {
    switch (a.rp) {
    case 1: secular_pair<1>(a, k, zq, uh, first_iter, viol_only, mvslot, dout); break;
    case 2: secular_pair<2>(a, k, zq, uh, first_iter, viol_only, mvslot, dout); break;
    case 3: secular_pair<3>(a, k, zq, uh, first_iter, viol_only, mvslot, dout); break;
    case 4: secular_pair<4>(a, k, zq, uh, first_iter, viol_only, mvslot, dout); break;
    case 5: secular_pair<5>(a, k, zq, uh, first_iter, viol_only, mvslot, dout); break;
    case 6: secular_pair<6>(a, k, zq, uh, first_iter, viol_only, mvslot, dout); break;
    case 7: secular_pair<7>(a, k, zq, uh, first_iter, viol_only, mvslot, dout); break;
    default: secular_pair<8>(a, k, zq, uh, first_iter, viol_only, mvslot, dout); break;
    }
}

__global__ __launch_bounds__(AF_THREADS) void admm_fused_kernel(AdmmFusedArgs a) {
    extern __shared__ double af_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> (cluster g, member c): the members of a cluster are 8 blocks apart, i.e. on one XCD under round-robin dispatch
    const int C = a.C;
    const int bx = (int)blockIdx.x & 7, brest = (int)blockIdx.x >> 3;
    const int c = brest % C, g = bx + 8 * (brest / C);
    if (g >= a.G) return;
    const int KBn = a.n16 / 16, MBh = a.Mh16 / 16, KSn = a.n16 / 4, KSh = a.Mh16 / 4;
    const int b_lo = (int)((int64_t)c * KBn / C), b_hi = (int)((int64_t)(c + 1) * KBn / C);
    const int NBl = b_hi - b_lo, rows = 16 * NBl, row0 = 16 * b_lo;
    const int k_lo = (int)((int64_t)c * a.m / C), k_hi = (int)((int64_t)(c + 1) * a.m / C);
    const int nk = k_hi - k_lo, rp = a.rp, h_lo = k_lo * rp, nh = nk * rp;
    const int rows_max = 16 * ((KBn + C - 1) / C), nh_max = ((a.m + C - 1) / C) * rp;
    // ---- LDS
    AfState &S = *reinterpret_cast<AfState *>(af_lds);
    double *Zs = af_lds + (sizeof(AfState) + 7) / 8;
    double *Ds = Zs + (size_t)rows_max * 16;
    double *UHs = Ds + (size_t)a.Mh16 * 16;
    double *ZQs = UHs + (size_t)nh_max * 16;
    const double dm = (double)a.m;

    for (int tile = g; tile < a.ntiles; tile += a.G) {
        double *Xt = a.X + (int64_t)tile * a.n16 * 16, *Bt = a.BEST + (int64_t)tile * a.n16 * 16;
        double *xb1 = a.xb1 + (int64_t)tile * C * (a.Mh16 + 2) * 16;       // [C][Mh16 + 2][16]
        double *xb1me = xb1 + (int64_t)c * (a.Mh16 + 2) * 16;
        double *xb2 = a.xb2 + (int64_t)tile * (a.Mh16 + C) * 16;           // [Mh16 + C][16]
        unsigned *fl1 = a.flags + ((int64_t)tile * 2 + 0) * C, *fl2 = a.flags + ((int64_t)tile * 2 + 1) * C;
        unsigned seq1 = 0, seq2 = 0;
        bool dead = false;       // abort seen (uniform over the workgroup)

        // publish: everything this member wrote for the exchange is ordered before the flag (barrier + release);
        // collect: wait for every member's flag (bounded), then everybody may read
        auto publish = [&](unsigned *fl, unsigned seq) {
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&fl[c], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        };
        auto collect = [&](const unsigned *fl, unsigned seq) {
            if (tid < C) {
                unsigned spins = 0;
                while ((int)(__hip_atomic_load(&fl[tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
                    if ((++spins & 1023u) == 0u &&
                        (spins > (1u << 24) || __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                        __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        S.abort = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            dead = S.abort != 0;
        };

        // ---- partial ZQ = W[rows, :]^T z[rows] -> this member's slot (hat blocks dealt to the waves, two at a time)
        auto gemm1 = [&]() {
            const int nks = 4 * NBl;
            for (int mb0 = wave; mb0 < MBh; mb0 += 16) {
                const int mb1 = mb0 + 8;
                const bool two = mb1 < MBh;
                const double *A0 = a.WTpk + ((int64_t)mb0 * KSn + 4 * b_lo) * 64 + lane;
                const double *A1 = a.WTpk + ((int64_t)(two ? mb1 : mb0) * KSn + 4 * b_lo) * 64 + lane;
                af_v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                double n0[4], n1[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { n0[u] = A0[u * 64]; n1[u] = A1[u * 64]; }
                for (int kk = 0; kk < nks; kk += 4) {
                    double c0[4], c1[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { c0[u] = n0[u]; c1[u] = n1[u]; }
                    if (kk + 4 < nks) {
#pragma unroll
                        for (int u = 0; u < 4; u++) { n0[u] = A0[(kk + 4 + u) * 64]; n1[u] = A1[(kk + 4 + u) * 64]; }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const double b = Zs[(kk + u) * 64 + lane];
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(c0[u], b, acc0, 0, 0, 0);
                        if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c1[u], b, acc1, 0, 0, 0);
                    }
                }
                // accumulator layout: register v of lane l = row (l >> 4) + 4 v of the block, column l & 15
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int hr = (lane >> 4) + 4 * v, col = lane & 15;
                    ag_store(xb1me + ((int64_t)(mb0 * 16 + hr)) * 16 + col, acc0[v]);
                    if (two) ag_store(xb1me + ((int64_t)(mb1 * 16 + hr)) * 16 + col, acc1[v]);
                }
            }
            if (tid < 32) ag_store(xb1me + ((int64_t)a.Mh16 + (tid >> 4)) * 16 + (tid & 15), (tid >> 4) ? S.pf0[tid & 15] : S.pd2[tid & 15]);
        };
        // ---- sums of the C partials for the own constraints' rows (fixed order), and of the two scalars per restart
        auto sum1 = [&]() {
            for (int idx = tid; idx < nh * 16; idx += AF_THREADS) {
                const int64_t off = ((int64_t)h_lo + (idx >> 4)) * 16 + (idx & 15);
                double s = ag_load(xb1 + off);
                for (int cc = 1; cc < C; cc++) s += ag_load(xb1 + (int64_t)cc * (a.Mh16 + 2) * 16 + off);
                ZQs[idx] = s;
            }
            if (tid < 32) {
                const int64_t off = ((int64_t)a.Mh16 + (tid >> 4)) * 16 + (tid & 15);
                double s = ag_load(xb1 + off);
                for (int cc = 1; cc < C; cc++) s += ag_load(xb1 + (int64_t)cc * (a.Mh16 + 2) * 16 + off);
                if (tid >> 4) S.f0z[tid & 15] = s + a.r0; else S.dist2[tid & 15] = s;
            }
            if (tid < 16) S.mvbits[tid] = 0ull;
            __syncthreads();
        };
        // ---- secular solves of the own (constraint, restart) pairs; publishes operand rows and partial max violations
        auto secular = [&](int first_iter, int viol_only) {
            for (int idx = tid; idx < nk * 16; idx += AF_THREADS) {
                const int kl = idx >> 4, r = idx & 15;
                if (!viol_only && !S.act[r]) continue;
                secular_dispatch(a, k_lo + kl, ZQs + (size_t)kl * rp * 16 + r, UHs + (size_t)kl * rp * 16 + r, first_iter, viol_only,
                                 &S.mvbits[r], xb2 + ((int64_t)(k_lo + kl) * rp) * 16 + r);
            }
            __syncthreads();
            if (tid < 16) ag_store(xb2 + ((int64_t)a.Mh16 + c) * 16 + tid, __longlong_as_double((long long)S.mvbits[tid]));
        };
        // ---- after exchange 2: all operand rows into LDS (unless only violations were wanted), max violation per restart
        auto gather2 = [&](int viol_only) {
            if (!viol_only)
                for (int idx = tid; idx < a.Mh16 * 16; idx += AF_THREADS) Ds[idx] = ag_load(xb2 + idx);
            if (tid < 16) {
                double mv = ag_load(xb2 + ((int64_t)a.Mh16) * 16 + tid);
                for (int cc = 1; cc < C; cc++) { const double w = ag_load(xb2 + ((int64_t)a.Mh16 + cc) * 16 + tid); mv = w > mv ? w : mv; }
                S.mvv[tid] = mv;
            }
            __syncthreads();
        };
        // ---- f0 of the z slice (partial over the own rows): sum (P0_ii z + q0) z, the expression of admm_f0_kernel
        auto f0_partial = [&]() {
            const int col = tid & 15, rl = tid >> 4;
            double acc = 0.0;
            for (int row = rl; row < rows; row += AF_THREADS / 16) {
                const int j = row0 + row;
                if (j < a.n) { const double z = Zs[row * 16 + col]; acc += (a.pdiag[j] * z + a.q0[j]) * z; }
            }
            S.scr[tid] = acc;
            __syncthreads();
            if (tid < 16) {
                double s = 0.0;
                for (int q = 0; q < AF_THREADS / 16; q++) s += S.scr[q * 16 + tid];
                S.pf0[tid] = s; S.pd2[tid] = 0.0;
            }
            __syncthreads();
        };
        // ---- (f0, max violation) of the point in Zs for all 16 restarts, in the arithmetic the iteration uses
        auto evaluate = [&](double *fout, double *vout) {
            f0_partial();
            gemm1();
            publish(fl1, ++seq1);
            collect(fl1, seq1);
            if (dead) return;
            sum1();
            secular(1, 1);
            publish(fl2, ++seq2);
            collect(fl2, seq2);
            if (dead) return;
            gather2(1);
            if (tid < 16) { fout[tid] = S.f0z[tid]; vout[tid] = S.mvv[tid]; }
            __syncthreads();
        };
        // ---- z-update of the own rows: T = W[rows, :] d on the matrix cores, element-wise update in the accumulators
        auto zupdate = [&](int phase) {
            double accd[4] = {0.0, 0.0, 0.0, 0.0}, accf[4] = {0.0, 0.0, 0.0, 0.0};   // per accumulator register, then per column
            const int col = lane & 15;
            const bool on = S.act[col] != 0;
            for (int lb0 = wave; lb0 < NBl; lb0 += 16) {
                const int lb1 = lb0 + 8;
                const bool two = lb1 < NBl;
                const double *A0 = a.Wpk + ((int64_t)(b_lo + lb0) * KSh) * 64 + lane;
                const double *A1 = a.Wpk + ((int64_t)(b_lo + (two ? lb1 : lb0)) * KSh) * 64 + lane;
                af_v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
                double n0[4], n1[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { n0[u] = A0[u * 64]; n1[u] = A1[u * 64]; }
                for (int kk = 0; kk < KSh; kk += 4) {
                    double c0[4], c1[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { c0[u] = n0[u]; c1[u] = n1[u]; }
                    if (kk + 4 < KSh) {
#pragma unroll
                        for (int u = 0; u < 4; u++) { n0[u] = A0[(kk + 4 + u) * 64]; n1[u] = A1[(kk + 4 + u) * 64]; }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const double b = Ds[(kk + u) * 64 + lane];
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(c0[u], b, acc0, 0, 0, 0);
                        if (two) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(c1[u], b, acc1, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    if (half && !two) break;
                    const int lb = half ? lb1 : lb0;
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const int row = lb * 16 + (lane >> 4) + 4 * v, j = row0 + row;
                        if (!on || j >= a.n) continue;
                        const double zold = Zs[row * 16 + col];
                        double s = half ? acc1[v] : acc0[v];
                        s += dm * zold;                                   // S = m z + W d  (reduced basis)
                        if (phase == 1) {
                            Zs[row * 16 + col] = s / dm;                  // qcqp.py:205
                        } else {
                            const double rhs = 2.0 * a.rho * s - a.q0[j]; // qcqp.py:231
                            const double zn = a.dinv[j] * rhs;
                            const double d = zold - zn;
                            accd[v] += d * d;
                            accf[v] += (a.pdiag[j] * zn + a.q0[j]) * zn;
                            Zs[row * 16 + col] = zn;
                        }
                    }
                }
            }
            if (phase == 2) {
                double sd = (accd[0] + accd[1]) + (accd[2] + accd[3]), sf = (accf[0] + accf[1]) + (accf[2] + accf[3]);
                sd += __shfl_xor(sd, 16); sd += __shfl_xor(sd, 32);
                sf += __shfl_xor(sf, 16); sf += __shfl_xor(sf, 32);
                if (lane < 16) { S.red[wave][0][lane] = sd; S.red[wave][1][lane] = sf; }
                __syncthreads();
                if (tid < 16) {
                    double d2 = 0.0, f = 0.0;
                    for (int w = 0; w < 8; w++) { d2 += S.red[w][0][tid]; f += S.red[w][1][tid]; }
                    S.pd2[tid] = d2; S.pf0[tid] = f;
                }
            }
            __syncthreads();
        };
        // ---- per-restart control flow (admm_book_kernel of admm.h), redundantly in every member
        auto book = [&](int phase, int have_last) {
            if (tid < 16) {
                int tk = 0, lv = 0;
                if (S.act[tid]) {
                    const double mv = S.mvv[tid];
                    bool stop = false;
                    if (phase == 1) {
                        if (mv < a.tol) stop = true;                                   // qcqp.py:203
                    } else {
                        if (have_last && sqrt(S.dist2[tid]) < a.tol) stop = true;      // qcqp.py:241-242 (before bestx)
                        else if (mv > a.viol_lim) stop = true;                         // qcqp.py:248
                        else if (better_first(S.f0z[tid], mv, S.best_f0[tid], S.best_mv[tid])) {
                            tk = 1; S.best_f0[tid] = S.f0z[tid]; S.best_mv[tid] = mv;  // bestx = better(z, bestx)
                        }
                    }
                    if (stop) S.act[tid] = 0;
                    else { if (phase == 1) S.it1[tid]++; else S.it2[tid]++; atomicAdd(&S.nactive, 1); }
                    lv = 1;
                }
                S.take[tid] = tk; S.live[tid] = lv;
            }
            __syncthreads();
            if (phase == 2) {
                const int col = tid & 15;
                if (S.take[col])
                    for (int row = tid >> 4; row < rows; row += AF_THREADS / 16) Bt[(int64_t)(row0 + row) * 16 + col] = Zs[row * 16 + col];
            }
        };
        auto run_phase = [&](int phase) {
            for (int idx = tid; idx < a.Mh16 * 16; idx += AF_THREADS) Ds[idx] = 0.0;     // xs = x0, us = 0: S = m x0
            if (tid < 16) { S.act[tid] = (tile * 16 + tid < a.R) ? 1 : 0; S.pd2[tid] = 0.0; S.pf0[tid] = 0.0; }
            __syncthreads();
            for (int t = 0; t < a.num_iters; t++) {
                zupdate(phase);
                gemm1();
                publish(fl1, ++seq1);
                collect(fl1, seq1);
                if (dead) return;
                sum1();
                secular(t == 0, 0);
                publish(fl2, ++seq2);
                collect(fl2, seq2);
                if (dead) return;
                if (tid == 0) S.nactive = 0;
                gather2(0);
                book(phase, t > 0);
                __syncthreads();
                if (S.nactive == 0) break;
                __syncthreads();
            }
        };

        // ================================================================ the run for this tile
        if (tid == 0) { S.abort = 0; S.nactive = 0; }
        if (tid < 16) { S.it1[tid] = 0; S.it2[tid] = 0; S.act[tid] = 0; }
        for (int idx = tid; idx < rows * 16; idx += AF_THREADS) {
            const int j = row0 + (idx >> 4);
            Zs[idx] = (j < a.n) ? Xt[(int64_t)j * 16 + (idx & 15)] : 0.0;
        }
        __syncthreads();
        evaluate(S.fx0, S.vx0);                                   // (f, v) of x0
        if (dead) return;
        if (tid < 16) { S.fx1[tid] = S.fx0[tid]; S.vx1[tid] = S.vx0[tid]; }
        __syncthreads();
        if (a.phase1) {
            run_phase(1);
            if (dead) return;
            __syncthreads();
            evaluate(S.f0z, S.mvv);                               // (f, v) of z1 (evaluate writes the arrays it is handed)
            if (dead) return;
            // x1 = better(x0, z1)  (qcqp.py:281)
            if (tid < 16) {
                const bool first = better_first(S.fx0[tid], S.vx0[tid], S.f0z[tid], S.mvv[tid]);
                S.take[tid] = first ? 1 : 0;
                if (!first) { S.fx1[tid] = S.f0z[tid]; S.vx1[tid] = S.mvv[tid]; }
            }
            __syncthreads();
            {
                const int col = tid & 15;
                for (int row = tid >> 4; row < rows; row += AF_THREADS / 16) {
                    const int64_t gi = (int64_t)(row0 + row) * 16 + col;
                    if (S.take[col]) Zs[row * 16 + col] = (row0 + row < a.n) ? Xt[gi] : 0.0;     // x0 stays
                    else Xt[gi] = Zs[row * 16 + col];                                           // X now holds x1
                }
            }
            __syncthreads();
        }
        // phase 2 from x1: bestx = x1
        for (int idx = tid; idx < rows * 16; idx += AF_THREADS) Bt[(int64_t)row0 * 16 + idx] = Zs[idx];
        if (tid < 16) { S.best_f0[tid] = S.fx1[tid]; S.best_mv[tid] = S.vx1[tid]; }
        __syncthreads();
        run_phase(2);
        if (dead) return;
        __syncthreads();
        // x2 = better(x1, bestx)  (qcqp.py:284)
        if (tid < 16) {
            const bool first = better_first(S.fx1[tid], S.vx1[tid], S.best_f0[tid], S.best_mv[tid]);
            S.take[tid] = first ? 1 : 0;
            const int64_t r = (int64_t)tile * 16 + tid;
            if (c == 0 && r < a.R) {
                a.f0_out[r] = first ? S.fx1[tid] : S.best_f0[tid];
                a.mv_out[r] = first ? S.vx1[tid] : S.best_mv[tid];
                a.iters1[r] = S.it1[tid];
                a.iters2[r] = S.it2[tid];
            }
        }
        __syncthreads();
        {
            const int col = tid & 15;
            if (!S.take[col])
                for (int row = tid >> 4; row < rows; row += AF_THREADS / 16) {
                    const int64_t gi = (int64_t)(row0 + row) * 16 + col;
                    Xt[gi] = Bt[gi];
                }
        }
        __syncthreads();
    }
}

}  // namespace

size_t admm_fused_lds_bytes(const AdmmFusedArgs &a) {
    const int KBn = a.n16 / 16;
    const size_t rows_max = 16 * (size_t)((KBn + a.C - 1) / a.C), nh_max = (size_t)((a.m + a.C - 1) / a.C) * a.rp;
    const size_t doubles = (sizeof(AfState) + 7) / 8 + rows_max * 16 + (size_t)a.Mh16 * 16 + 2 * nh_max * 16;
    const size_t bytes = doubles * sizeof(double);
    return bytes <= 160 * 1024 ? bytes : 0;
}

int admm_fused_max_clusters(const AdmmFusedArgs &a, int device) {
    const size_t lds = admm_fused_lds_bytes(a);
    if (!lds) return 0;
    (void)hipFuncSetAttribute((const void *)admm_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, admm_fused_kernel, AF_THREADS, lds) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    const int blocks = per_cu * cus;
    // block index = x + 8 (c + C y): clusters come in groups of 8
    int G = (blocks / (8 * a.C)) * 8;
    return G;
}

int admm_fused_launch(const AdmmFusedArgs &a, hipStream_t st) {
    const size_t lds = admm_fused_lds_bytes(a);
    if (!lds) return (int)hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute((const void *)admm_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const int groups = (a.G + 7) / 8;
    AdmmFusedArgs args = a;
    void *params[] = {&args};
    e = hipLaunchCooperativeKernel((const void *)admm_fused_kernel, dim3((unsigned)(8 * a.C * groups)), dim3(AF_THREADS), params, (unsigned)lds, st);
    return (int)e;
}

}  // namespace qcqpmi
