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
typedef __attribute__((address_space(1))) const double af_gcd;     // global memory, read-only streams (global_load, not flat)

__device__ inline double ag_load(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void ag_store(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __attribute__((always_inline)) inline int af_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }   // wave-uniform -> SGPR

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

// acc[i] += sum_k A[i][k] B[k], i < NA, over nks k-steps (a multiple of 4): A fragments stream from global memory (L2)
// 64 doubles apart, B operands from LDS 64 doubles apart (one LDS read serves NA MFMAs).  Groups of 4 k-steps through
// THREE register stages named statically (the loop is unrolled by three groups: no register rotation, hence no wait on a
// load issued one group ago): while group i multiplies, the operands of group i + 2 are requested.  The scheduling
// barriers keep the compiler from sinking the requests next to their uses.
constexpr int AF_NA = 3;      // accumulators (blocks of 16 rows) a multiplying wave carries at once

#ifndef AF_NS
#define AF_NS 3          // register stages of the fragment stream (AF_NS - 1 groups of requests in flight)
#endif
#ifdef AF_SAMEFRAG       // experiment (results invalid): every k-step reads the first fragment, i.e. the stream comes from the L1
#define AF_FRAG(k) 0
#else
#define AF_FRAG(k) ((k) * 64)
#endif

template <int NA>
__device__ __attribute__((always_inline)) inline void af_stream(af_gcd *const (&A)[AF_NA], const double *Bs, int nks, af_v4d (&acc)[AF_NA]) {
    constexpr int GS = NA >= 3 ? 2 : 4;          // k-steps per group: 8 MFMAs either way (6 for three accumulators)
    constexpr int NS = AF_NS;
    double r[NS][NA][GS], rb[NS][GS];
    const int last = nks - GS;
#pragma unroll
    for (int j = 0; j < NS - 1; j++) {
        const int k = GS * j < nks ? GS * j : last;
#pragma unroll
        for (int u = 0; u < GS; u++) {
#pragma unroll
            for (int i = 0; i < NA; i++) r[j][i][u] = A[i][AF_FRAG(k + u)];
            rb[j][u] = Bs[(k + u) * 64];
        }
        // keeps the groups of requests in program order: the compiler issued group 1 BEFORE group 0, its wait-count
        // bookkeeping then held group 0 to be the youngest at the loop header and made the first group of every pass wait for
        // ALL requests but its own (s_waitcnt vmcnt(8) instead of vmcnt(16): one group of lead instead of two)
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int kk = 0; kk < nks; kk += NS * GS) {
#pragma unroll
        for (int j = 0; j < NS; j++) {
            const int kg = kk + GS * j;
            const int k2 = kg + (NS - 1) * GS < nks ? kg + (NS - 1) * GS : last;     // clamped: requested, never multiplied
            const int s2 = (j + NS - 1) % NS;
#pragma unroll
            for (int u = 0; u < GS; u++) {
#pragma unroll
                for (int i = 0; i < NA; i++) r[s2][i][u] = A[i][AF_FRAG(k2 + u)];
                rb[s2][u] = Bs[(k2 + u) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (kg < nks) {      // wave-uniform
#pragma unroll
                for (int u = 0; u < GS; u++)
#pragma unroll
                    for (int i = 0; i < NA; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(r[j][i][u], rb[j][u], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__device__ __attribute__((always_inline)) inline void af_stream_n(int na, af_gcd *const (&A)[AF_NA], const double *Bs, int nks, af_v4d (&acc)[AF_NA]) {
    if (na >= 3) af_stream<3>(A, Bs, nks, acc);
    else if (na == 2) af_stream<2>(A, Bs, nks, acc);
    else af_stream<1>(A, Bs, nks, acc);
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

// onecons_qcqp on the RP coordinates of a reduced basis for ONE (constraint, restart) pair: admm_secular_small_kernel of
// admm.h with the operands in LDS.  zq / uh: column of the pair, rows 16 doubles apart; dout: the same in global memory.
// QZ: every qhat is zero (constraints without linear terms, the beamforming family): -(nu 0 - 2 v) = 2 v and + 0 xh
// drop out of the secular function bit for bit.
template <int RP, bool QZ>
__device__ __attribute__((always_inline)) inline void secular_pair(const AdmmFusedArgs &a, int k, const double *zq, double *uh, bool first_iter,
                                                                   bool viol_only, unsigned long long *mvslot, double *dout) {
    af_gcd *lm = (af_gcd *)a.lam + (int64_t)k * RP, *qh = (af_gcd *)a.qhat + (int64_t)k * RP;
    const double rk = ((af_gcd *)a.rk)[k];
    const int relop = ((__attribute__((address_space(1))) const int *)a.relop)[k];
    double L[RP], Qh[RP], V[RP], V2[RP], Zq[RP], X[RP];
    double fz = 0.0, fv = 0.0;
#pragma unroll
    for (int e = 0; e < RP; e++) {
        L[e] = lm[e]; Qh[e] = QZ ? 0.0 : qh[e];
        Zq[e] = zq[e * 16];
        const double u = (!first_iter && !viol_only) ? uh[e * 16] : 0.0;
        V[e] = Zq[e] + u;
        V2[e] = 2.0 * V[e];
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
                const double num = QZ ? V2[e] : -(nu * Qh[e] - 2.0 * V[e]);
                const double xh = (L[e] != 0.0) ? af_div(num, 2.0 * (1.0 + nu * L[e])) : num * 0.5;
                X[e] = xh;
                if (QZ) p += L[e] * (xh * xh); else p += L[e] * (xh * xh) + Qh[e] * xh;
            }
            return p + rk;
        };
        double s = ((af_gcd *)a.slo)[k], e_ = ((af_gcd *)a.ehi)[k];
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

enum { AF_EVAL0 = 0, AF_PH1 = 1, AF_EVAL1 = 2, AF_PH2 = 3, AF_DONE = 4 };

// One instance of every stage inside one loop that walks EVAL0 -> PH1 iterations -> EVAL1 -> PH2 iterations (the first
// version instantiated the stages per call site: 120 KB of code against a 64 KB instruction cache).
template <int RP, int NT>
__global__ __launch_bounds__(NT, 2) void admm_fused_kernel(AdmmFusedArgs a) {
    constexpr int MW = NT / 64;          // every wave multiplies

    extern __shared__ double af_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = af_uni(tid >> 6);
    // block -> (cluster g, member c): the members of a cluster are 8 blocks apart, i.e. on one XCD under round-robin dispatch
    const int C = a.C;
    const int bx = (int)blockIdx.x & 7, brest = (int)blockIdx.x >> 3;
    const int c = af_uni(brest % C), g = af_uni(bx + 8 * (brest / C));
    if (g >= a.G) return;
    const int KBn = a.n16 / 16, MBh = a.Mh16 / 16, KSn = a.n16 / 4, KSh = a.Mh16 / 4;
    const int b_lo = af_uni((int)((int64_t)c * KBn / C)), b_hi = af_uni((int)((int64_t)(c + 1) * KBn / C));
    const int NBl = b_hi - b_lo, rows = 16 * NBl, row0 = 16 * b_lo;
    const int k_lo = af_uni((int)((int64_t)c * a.m / C)), k_hi = af_uni((int)((int64_t)(c + 1) * a.m / C));
    const int nk = k_hi - k_lo, h_lo = k_lo * RP, nh = nk * RP;
    const int rows_max = af_uni(16 * ((KBn + C - 1) / C)), nh_max = af_uni(((a.m + C - 1) / C) * RP);
    // ---- LDS
    AfState &S = *reinterpret_cast<AfState *>(af_lds);
    double *Zs = af_lds + (sizeof(AfState) + 7) / 8;
    double *Ds = Zs + (size_t)rows_max * 16;
    double *UHs = Ds + (size_t)a.Mh16 * 16;
    double *ZQs = UHs + (size_t)nh_max * 16;
    double *Qs = ZQs + (size_t)nh_max * 16, *DIs = Qs + rows_max, *PDs = DIs + rows_max;   // q0, 1/(2(P0_ii + rho m)), P0_ii of the own rows
    const double dm = (double)a.m;
    af_gcd *WT = (af_gcd *)a.WTpk, *Wp = (af_gcd *)a.Wpk;
    const bool qz = a.qzero != 0;

    for (int tile = g; tile < a.ntiles; tile += a.G) {
        double *Xt = a.X + (int64_t)tile * a.n16 * 16, *Bt = a.BEST + (int64_t)tile * a.n16 * 16;
        double *xb1 = a.xb1 + (int64_t)tile * C * (a.Mh16 + 2) * 16;       // [C][Mh16 + 2][16]
        double *xb1me = xb1 + (int64_t)c * (a.Mh16 + 2) * 16;
        double *xb2 = a.xb2 + (int64_t)tile * (a.Mh16 + C) * 16;           // [Mh16 + C][16]
        unsigned *fl1 = a.flags + ((int64_t)tile * 2 + 0) * C, *fl2 = a.flags + ((int64_t)tile * 2 + 1) * C;
        unsigned seq = 0;
        const bool profiling = a.prof != nullptr && tile == 0 && c == 0 && tid == 0;
        long long pt = profiling ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define AF_TICK(slot) if (profiling) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); a.prof[slot] += now_ - pt; pt = now_; }

        // Exchanges.  Every shared word is accessed with agent-scope (sc1) atomics, which are coherent across the XCDs on
        // their own; what a release / acquire pair would add -- buffer_wbl2 (write back the whole L2) per publish and
        // buffer_inv (drop the L2's lines, the streamed W fragments among them) per poll -- is not needed, only the ORDER
        // "data complete, then flag": every thread drains its stores (s_waitcnt vmcnt(0): an sc1 store is acknowledged at
        // device scope), the barrier collects the threads, one relaxed store raises the flag.  The reader polls with relaxed
        // loads and issues its data loads after the barrier that follows the poll (loads of a wave return in order).
#define AF_EXCHANGE(fl)                                                                                                         \
        {                                                                                                                       \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                    \
            __syncthreads();                                                                                                    \
            if (tid == 0) __hip_atomic_store(&(fl)[c], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                        \
            if (tid < C) {                                                                                                      \
                unsigned spins = 0;                                                                                             \
                while ((int)(__hip_atomic_load(&(fl)[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {           \
                    if ((++spins & 1023u) == 0u &&                                                                              \
                        (spins > (1u << 24) || __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {  \
                        __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                        \
                        S.abort = 1;                                                                                            \
                        break;                                                                                                  \
                    }                                                                                                           \
                    __builtin_amdgcn_s_sleep(1);                                                                                \
                }                                                                                                               \
            }                                                                                                                   \
            asm volatile("" ::: "memory");                                                                                      \
            __syncthreads();                                                                                                    \
            if (af_uni(S.abort)) return;                                                                                        \
        }

        // ================================================================ set-up of the tile
        if (tid == 0) { S.abort = 0; S.nactive = 0; }
        if (tid < 16) { S.it1[tid] = 0; S.it2[tid] = 0; S.act[tid] = 0; S.pd2[tid] = 0.0; S.pf0[tid] = 0.0; }
        for (int idx = tid; idx < rows * 16; idx += NT) {
            const int j = row0 + (idx >> 4);
            Zs[idx] = (j < a.n) ? Xt[(int64_t)j * 16 + (idx & 15)] : 0.0;
        }
        for (int row = tid; row < rows; row += NT) {
            const int j = row0 + row;
            Qs[row] = j < a.n ? a.q0[j] : 0.0; DIs[row] = j < a.n ? a.dinv[j] : 0.0; PDs[row] = j < a.n ? a.pdiag[j] : 0.0;
        }
        __syncthreads();

        int st = AF_EVAL0, t = 0;
        while (st != AF_DONE) {
            const bool iter = st == AF_PH1 || st == AF_PH2;
            const bool ph2 = st == AF_PH2;
            AF_TICK(8)
            if (iter) {
                // ---- z-update of the own rows: T = W[rows, :] d on the matrix cores, element-wise in the accumulators
                double accd = 0.0, accf = 0.0;
                const int col = lane & 15;
                const bool on = S.act[col] != 0;
                for (int lb0 = wave; wave < MW && lb0 < NBl; lb0 += MW * AF_NA) {
                    int na = 0;
                    af_gcd *A[AF_NA];
                    af_v4d acc[AF_NA];
#pragma unroll
                    for (int i = 0; i < AF_NA; i++) {
                        const int lb = lb0 + MW * i;
                        if (lb < NBl) na = i + 1;
                        A[i] = Wp + ((int64_t)(b_lo + (lb < NBl ? lb : lb0)) * KSh) * 64 + lane;
                        acc[i] = af_v4d{0.0, 0.0, 0.0, 0.0};
                    }
                    const long long ps0 = profiling ? (long long)__builtin_amdgcn_s_memtime() : 0;
                    af_stream_n(na, A, Ds + lane, KSh, acc);
                    if (profiling) a.prof[10] += (long long)__builtin_amdgcn_s_memtime() - ps0;
                    // element-wise part, free of branches: the operands of a block's four rows are requested together (the
                    // version with a `continue` per row ran the rows one LDS round trip after the other: 4.6 k cycles per call)
#pragma unroll
                    for (int i = 0; i < AF_NA; i++) {
                        if (i >= na) break;
                        const int lb = lb0 + MW * i;
                        double zo[4], qs[4], di[4], pd[4];
#pragma unroll
                        for (int v = 0; v < 4; v++) {
                            const int row = lb * 16 + (lane >> 4) + 4 * v;
                            zo[v] = Zs[row * 16 + col];
                            qs[v] = Qs[row]; di[v] = DIs[row]; pd[v] = PDs[row];
                        }
#pragma unroll
                        for (int v = 0; v < 4; v++) {
                            const int row = lb * 16 + (lane >> 4) + 4 * v;
                            const bool ok = on && row0 + row < a.n;
                            double sm = acc[i][v];
                            sm += dm * zo[v];                                    // S = m z + W d  (reduced basis)
                            double zn;
                            if (!ph2) {
                                zn = sm / dm;                                    // qcqp.py:205
                            } else {
                                const double rhs = 2.0 * a.rho * sm - qs[v];     // qcqp.py:231
                                zn = di[v] * rhs;
                                const double d = zo[v] - zn;
                                accd += ok ? d * d : 0.0;
                                accf += ok ? (pd[v] * zn + qs[v]) * zn : 0.0;
                            }
                            if (ok) Zs[row * 16 + col] = zn;
                        }
                    }
                }
                if (profiling) a.prof[12] += (long long)__builtin_amdgcn_s_memtime() - pt;     // wave 0: product + element-wise part
                if (ph2) {
                    accd += __shfl_xor(accd, 16); accd += __shfl_xor(accd, 32);
                    accf += __shfl_xor(accf, 16); accf += __shfl_xor(accf, 32);
                    if (lane < 16) { S.red[wave][0][lane] = accd; S.red[wave][1][lane] = accf; }
                    __syncthreads();
                    if (tid < 16) {
                        double d2 = 0.0, f = 0.0;
                        for (int w = 0; w < MW; w++) { d2 += S.red[w][0][tid]; f += S.red[w][1][tid]; }
                        S.pd2[tid] = d2; S.pf0[tid] = f;
                    }
                }
                __syncthreads();
            } else {
                // ---- evaluation of the point in Zs: f0 partial over the own rows, sum (P0_ii z + q0) z (admm_f0_kernel)
                const int col = tid & 15;
                double acc = 0.0;
                for (int row = tid >> 4; row < rows; row += NT / 16)
                    if (row0 + row < a.n) { const double z = Zs[row * 16 + col]; acc += (PDs[row] * z + Qs[row]) * z; }
                S.scr[tid] = acc;
                __syncthreads();
                if (tid < 16) {
                    double s = 0.0;
                    for (int q = 0; q < NT / 16; q++) s += S.scr[q * 16 + tid];
                    S.pf0[tid] = s; S.pd2[tid] = 0.0;
                }
                __syncthreads();
            }
            AF_TICK(0)
            // ---- partial ZQ = W[rows, :]^T z[rows] -> this member's slot (hat blocks dealt to the waves, two at a time)
            for (int mb0 = wave; wave < MW && mb0 < MBh; mb0 += MW * AF_NA) {
                int na = 0;
                af_gcd *A[AF_NA];
                af_v4d acc[AF_NA];
#pragma unroll
                for (int i = 0; i < AF_NA; i++) {
                    const int mb = mb0 + MW * i;
                    if (mb < MBh) na = i + 1;
                    A[i] = WT + ((int64_t)(mb < MBh ? mb : mb0) * KSn + 4 * b_lo) * 64 + lane;
                    acc[i] = af_v4d{0.0, 0.0, 0.0, 0.0};
                }
                const long long ps0 = profiling ? (long long)__builtin_amdgcn_s_memtime() : 0;
                af_stream_n(na, A, Zs + lane, 4 * NBl, acc);
                if (profiling) a.prof[11] += (long long)__builtin_amdgcn_s_memtime() - ps0;
                // accumulator layout: register v of lane l = row (l >> 4) + 4 v of the block, column l & 15
#pragma unroll
                for (int i = 0; i < AF_NA; i++) {
                    if (i >= na) break;
                    const int mb = mb0 + MW * i;
#pragma unroll
                    for (int v = 0; v < 4; v++) ag_store(xb1me + ((int64_t)(mb * 16 + (lane >> 4) + 4 * v)) * 16 + (lane & 15), acc[i][v]);
                }
            }
            if (tid < 32) ag_store(xb1me + ((int64_t)a.Mh16 + (tid >> 4)) * 16 + (tid & 15), (tid >> 4) ? S.pf0[tid & 15] : S.pd2[tid & 15]);
            AF_TICK(1)
            ++seq;
            AF_EXCHANGE(fl1)
            AF_TICK(2)
            // ---- sums of the C partials for the own constraints' rows (fixed order), and of the two scalars per restart (two
            // more rows of the same pass).  Two items per thread and pass, all 2 C loads of a thread in flight before the first
            // sum waits: ONE round trip to the exchange buffer for up to 2 NT items (round 5; it was one per NT items plus one
            // for the scalars: 3 round trips with four-wave workgroups)
            {
                const int64_t slot = (int64_t)(a.Mh16 + 2) * 16;
                const int nitem = nh * 16 + 32;
                for (int i0 = tid; i0 < nitem; i0 += 2 * NT) {
                    double pv[2][AF_MAXC];
                    int64_t off[2];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int i = i0 + u * NT;
                        const bool in = i < nitem, sc = i >= nh * 16;
                        const int j = sc ? i - nh * 16 : i;
                        off[u] = in ? ((int64_t)(sc ? a.Mh16 : h_lo) + (j >> 4)) * 16 + (j & 15) : 0;
#pragma unroll
                        for (int cc = 0; cc < AF_MAXC; cc++) pv[u][cc] = (in && cc < C) ? ag_load(xb1 + cc * slot + off[u]) : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int i = i0 + u * NT;
                        if (i >= nitem) break;
                        double sum = pv[u][0];
#pragma unroll
                        for (int cc = 1; cc < AF_MAXC; cc++) if (cc < C) sum += pv[u][cc];
                        if (i < nh * 16) ZQs[i] = sum;
                        else { const int j = i - nh * 16; if (j >> 4) S.f0z[j & 15] = sum + a.r0; else S.dist2[j & 15] = sum; }
                    }
                }
                if (tid < 16) S.mvbits[tid] = 0ull;
                __syncthreads();
            }
            AF_TICK(3)
            // ---- secular solves of the own (constraint, restart) pairs; operand rows and partial max violations go out
            {
                const bool first_iter = !iter || t == 0, viol_only = !iter;
                for (int idx = tid; idx < nk * 16; idx += NT) {
                    const int kl = idx >> 4, r = idx & 15;
                    if (!viol_only && !S.act[r]) continue;
                    const double *zq = ZQs + (size_t)kl * RP * 16 + r;
                    double *uh = UHs + (size_t)kl * RP * 16 + r, *dout = xb2 + ((int64_t)(k_lo + kl) * RP) * 16 + r;
                    if (qz) secular_pair<RP, true>(a, k_lo + kl, zq, uh, first_iter, viol_only, &S.mvbits[r], dout);
                    else secular_pair<RP, false>(a, k_lo + kl, zq, uh, first_iter, viol_only, &S.mvbits[r], dout);
                }
                __syncthreads();
                if (tid < 16) ag_store(xb2 + ((int64_t)a.Mh16 + c) * 16 + tid, __longlong_as_double((long long)S.mvbits[tid]));
            }
            AF_TICK(4)
            AF_EXCHANGE(fl2)
            AF_TICK(5)
            // ---- all operand rows into LDS (iterations only) and the members' partial max violations: every load of a thread
            // in flight at once (GU per pass: one round trip for Mh16 <= GU NT / 16 rows; it was four loads per pass)
            {
                constexpr int GU = NT == 256 ? 10 : 6;
                if (tid == 0) S.nactive = 0;
                const int nd = iter ? a.Mh16 * 16 : 0, ng = nd + C * 16;
                for (int idx0 = 0; idx0 < ng; idx0 += GU * NT) {
                    double pv[GU];
#pragma unroll
                    for (int u = 0; u < GU; u++) {
                        const int idx = idx0 + u * NT + tid;
                        pv[u] = (idx < ng) ? ag_load(xb2 + (idx < nd ? idx : a.Mh16 * 16 + (idx - nd))) : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < GU; u++) {
                        const int idx = idx0 + u * NT + tid;
                        if (idx < nd) Ds[idx] = pv[u];
                        else if (idx < ng) S.scr[idx - nd] = pv[u];
                    }
                }
                __syncthreads();
                if (tid < 16) {
                    double mv = S.scr[tid];
                    for (int cc = 1; cc < C; cc++) { const double w = S.scr[cc * 16 + tid]; mv = w > mv ? w : mv; }
                    S.mvv[tid] = mv;               // read below by the same thread
                }
            }
            AF_TICK(6)
            int next = st;
            if (iter) {
                // ---- per-restart control flow (admm_book_kernel of admm.h), redundantly in every member
                if (tid < 16) {
                    int tk = 0;
                    if (S.act[tid]) {
                        const double mv = S.mvv[tid];
                        bool stop = false;
                        if (!ph2) {
                            if (mv < a.tol) stop = true;                                   // qcqp.py:203
                        } else {
                            if (t > 0 && sqrt(S.dist2[tid]) < a.tol) stop = true;          // qcqp.py:241-242 (before bestx)
                            else if (mv > a.viol_lim) stop = true;                         // qcqp.py:248
                            else if (better_first(S.f0z[tid], mv, S.best_f0[tid], S.best_mv[tid])) {
                                tk = 1; S.best_f0[tid] = S.f0z[tid]; S.best_mv[tid] = mv;  // bestx = better(z, bestx)
                            }
                        }
                        if (stop) S.act[tid] = 0;
                        else { if (!ph2) S.it1[tid]++; else S.it2[tid]++; atomicAdd(&S.nactive, 1); }
                    }
                    S.take[tid] = tk;
                }
                __syncthreads();
                if (ph2) {
                    const int col = tid & 15;
                    if (S.take[col])
                        for (int row = tid >> 4; row < rows; row += NT / 16) Bt[(int64_t)(row0 + row) * 16 + col] = Zs[row * 16 + col];
                }
                t++;
                if (profiling) a.prof[9]++;
                if (af_uni(S.nactive) == 0 || t >= a.num_iters) next = ph2 ? AF_DONE : AF_EVAL1;
            } else {
                // ---- an evaluation has finished: (f, v) of the point in Zs are in f0z / mvv
                if (st == AF_EVAL0) {
                    if (tid < 16) { S.fx0[tid] = S.f0z[tid]; S.vx0[tid] = S.mvv[tid]; S.fx1[tid] = S.f0z[tid]; S.vx1[tid] = S.mvv[tid]; }
                    next = (a.phase1 && a.num_iters > 0) ? AF_PH1 : AF_PH2;
                } else {
                    // x1 = better(x0, z1)  (qcqp.py:281)
                    if (tid < 16) {
                        const bool first = better_first(S.fx0[tid], S.vx0[tid], S.f0z[tid], S.mvv[tid]);
                        S.take[tid] = first ? 1 : 0;
                        if (!first) { S.fx1[tid] = S.f0z[tid]; S.vx1[tid] = S.mvv[tid]; }
                    }
                    __syncthreads();
                    const int col = tid & 15;
                    for (int row = tid >> 4; row < rows; row += NT / 16) {
                        const int64_t gi = (int64_t)(row0 + row) * 16 + col;
                        if (S.take[col]) Zs[row * 16 + col] = (row0 + row < a.n) ? Xt[gi] : 0.0;     // x0 stays
                        else Xt[gi] = Zs[row * 16 + col];                                           // X now holds x1
                    }
                    next = AF_PH2;
                }
                __syncthreads();
            }
            if (next != st && (next == AF_PH1 || next == AF_PH2)) {
                // ---- start of a phase: xs = x0, us = 0 (S = m x0: no operand rows yet); phase 2 starts with bestx = x1
                for (int idx = tid; idx < a.Mh16 * 16; idx += NT) Ds[idx] = 0.0;
                if (tid < 16) { S.act[tid] = (tile * 16 + tid < a.R) ? 1 : 0; S.pd2[tid] = 0.0; S.pf0[tid] = 0.0; }
                if (next == AF_PH2) {
                    for (int idx = tid; idx < rows * 16; idx += NT) Bt[(int64_t)row0 * 16 + idx] = Zs[idx];
                    if (tid < 16) { S.best_f0[tid] = S.fx1[tid]; S.best_mv[tid] = S.vx1[tid]; }
                    if (a.num_iters <= 0) next = AF_DONE;
                }
                t = 0;
            }
            __syncthreads();
            AF_TICK(7)
            st = next;
        }
        // ================================================================ x2 = better(x1, bestx)  (qcqp.py:284)
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
                for (int row = tid >> 4; row < rows; row += NT / 16) {
                    const int64_t gi = (int64_t)(row0 + row) * 16 + col;
                    Xt[gi] = Bt[gi];
                }
        }
        __syncthreads();
#undef AF_TICK
#undef AF_EXCHANGE
    }
}

typedef void (*af_kernel_t)(AdmmFusedArgs);
template <int NT>
af_kernel_t af_kernel_nt(int rp) {
    switch (rp) {
    case 1: return admm_fused_kernel<1, NT>;
    case 2: return admm_fused_kernel<2, NT>;
    case 3: return admm_fused_kernel<3, NT>;
    case 4: return admm_fused_kernel<4, NT>;
    case 5: return admm_fused_kernel<5, NT>;
    case 6: return admm_fused_kernel<6, NT>;
    case 7: return admm_fused_kernel<7, NT>;
    case 8: return admm_fused_kernel<8, NT>;
    }
    return nullptr;
}
af_kernel_t af_kernel_for(int rp, int nt) { return nt == 256 ? af_kernel_nt<256>(rp) : nt == 512 ? af_kernel_nt<512>(rp) : nullptr; }

}  // namespace

size_t admm_fused_lds_bytes(const AdmmFusedArgs &a) {
    const int KBn = a.n16 / 16;
    const size_t rows_max = 16 * (size_t)((KBn + a.C - 1) / a.C), nh_max = (size_t)((a.m + a.C - 1) / a.C) * a.rp;
    const size_t doubles = (sizeof(AfState) + 7) / 8 + rows_max * 16 + (size_t)a.Mh16 * 16 + 2 * nh_max * 16 + 3 * rows_max;
    const size_t bytes = doubles * sizeof(double);
    return bytes <= 160 * 1024 ? bytes : 0;
}

int admm_fused_max_clusters(const AdmmFusedArgs &a, int device) {
    const size_t lds = admm_fused_lds_bytes(a);
    if (!lds || !af_kernel_for(a.rp, a.nt)) return 0;
    (void)hipFuncSetAttribute((const void *)af_kernel_for(a.rp, a.nt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, af_kernel_for(a.rp, a.nt), a.nt, lds) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    const int blocks = per_cu * cus;
    // block index = x + 8 (c + C y): clusters come in groups of 8
    int G = (blocks / (8 * a.C)) * 8;
    return G;
}

int admm_fused_launch(const AdmmFusedArgs &a, hipStream_t st) {
    const size_t lds = admm_fused_lds_bytes(a);
    if (!lds || !af_kernel_for(a.rp, a.nt)) return (int)hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute((const void *)af_kernel_for(a.rp, a.nt), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const int groups = (a.G + 7) / 8;
    AdmmFusedArgs args = a;
    void *params[] = {&args};
    e = hipLaunchCooperativeKernel((const void *)af_kernel_for(a.rp, a.nt), dim3((unsigned)(8 * a.C * groups)), dim3((unsigned)a.nt), params, (unsigned)lds, st);
    return (int)e;
}

}  // namespace qcqpmi
