// dense_chain_mw_kernel -- the chain of the dense-constraint path (cd_dense.h) with ONE WORKGROUP OF FOUR WAVES PER RESTART
// (up to eight; round 3).  dense_chain_kernel gives a restart one wavefront: with m = 256 constraints every lane walks 5 function slots
// per coordinate (coefficients, two square roots and five divisions per slot through the divergent branches of the interval
// rules, ...) and a block of 16 coordinates costs 200 us whatever the number of restarts.  Here the functions are dealt to
// the threads of up to eight waves (one function per thread up to 512 functions: 5 waves at m = 256; 3 slots per thread on
// 6 waves at m = 1024) and everything a thread needs for its slots
// stays in REGISTERS between the stages of a coordinate; LDS holds the tracked function values, the gap / segment lists and
// a few words of reductions and broadcasts -- 5 KB per restart at m = 256 instead of 60 KB, so the chain's workgroups share
// the CUs with the products kernel of the next block (64 KB per workgroup) instead of waiting for each other.
//
// Per coordinate (utilities.py:99-105, 209-288; qcqp.py:101-178):
//   A. every thread requests what its slots need in one batch -- the row of G (K-split partials), the diagonal entry, the
//      linear coefficient and the entries of the diagonal block for the moves already made in this block (Gauss-Seidel) --
//      and forms (t2, t1, t0): same expressions and the same order of operations as dense_chain_kernel, bit for bit;
//   B. bounds of the own constraints, wave reductions (DPP), one LDS word per wave, barrier, combined by every thread in
//      the same order; gaps that cut into [L, H] are appended to the list with an LDS atomic (the list is sorted before it
//      is swept, equal starts are merged as a group: the order of arrival cannot matter); barrier;
//   C. the serial thread sweeps the segments and minimises / draws (keyed Philox: the draw does not depend on who makes it), decides
//      and publishes the move; barrier;
//   D. every thread commits its tracked function values.
// Phase 1 runs the reference's bisection on the slack around B-C.
#pragma once
#include "cd_dense.h"

namespace qcqpmi {

constexpr int MW_W = 8;                                                // at most this many waves per restart
constexpr int MW_TMAX = 64 * MW_W;
constexpr int MW_LDS_FIXED = 2 * DN_GC + 2 * DN_SC + 32 + 4 * MW_W + 8 + 8;   // doubles besides F[m1p]

// Geometry for m = m1 - 1 constraints.  Constraint k = 1, 2, ... is slot j = (k - 1) / Tc of thread (k - 1) % Tc, where Tc
// (whole waves) is the number of threads that hold constraints.  The objective (k = 0) has no interval to evaluate; it is
// slot 0 of the SERIAL thread ts -- the first lane that holds no constraint (a free lane of the last wave when there is one,
// else lane 0 of an extra wave) -- which also sweeps the segments and minimises: m = 256 is four full waves of interval
// arithmetic plus a fifth wave that only runs the serial part.
struct MwGeom { int SL, Tc, ts, T; };
inline MwGeom mw_geometry(int m1) {
    const int m = m1 - 1, nwc = m > 0 ? (m + 63) / 64 : 1;
    MwGeom g;
    for (g.SL = 1;; g.SL++) {
        g.Tc = 64 * ((nwc + g.SL - 1) / g.SL);
        g.ts = (g.SL == 1 && m < g.Tc) ? m : g.Tc;
        g.T = 64 * (g.ts / 64 + 1);
        if (g.T <= MW_TMAX) break;
    }
    return g;
}

struct MwLds {
    double *F;                     // [m1p] tracked function values
    double *gapa, *gapb;           // [DN_GC]
    double *seglo, *seghi;         // [DN_SC]
    double *xb, *dlt;              // [16]
    unsigned long long *kL, *kH, *kV;   // reductions over the threads: order-preserving keys of max lo, min hi, max violation (LDS atomics)
    int *red;                      // [8] 0 multiplicity of H, 1 an empty constraint was seen, 2 an involved constraint was seen
    double *bx;                    // [2] broadcast: the point
    int *bi;                       // [8] 0 segments, 1 got, 2 unbounded, 3 gap count (atomic), 4 overflow, 5 moved
};

// p x^2 + q x + rs <= 0: intervals_le (onevar.h) without its branches
__device__ inline Seg2 mw_intervals_le(double p, double q, double rs, double smr) {
    // Straight-line form: ONE square root and TWO divisions whatever mix of convex, concave and linear functions the lanes of
    // a wave hold; every case selects the operands of the reference's expression for it (same operations on the same values,
    // bit for bit), results that a case does not use are discarded.
    const double tol = 1e-4;
    const bool pos = p > tol, neg = p < -tol, quad = pos || neg;
    const double D = q * q - 4.0 * p * rs;
    const bool real = D >= 0.0;
    const double rD = sqrt(real ? D : 0.0);
    const double den = quad ? 2.0 * p : q;
    const double a = (quad ? (-q - rD) : smr) / den;        // (-q - rD) / (2 p)   or   (s - r) / q
    const double b = (-q + rD) / den;                       // (-q + rD) / (2 p)
    const bool qp = q > tol, qn = q < -tol;
    Seg2 o;
    // number of intervals: convex 1 (0 without real roots); concave 2 (1 = the whole line without real roots); linear 1
    o.n = quad ? (real ? (pos ? 1 : 2) : (pos ? 0 : 1)) : 1;
    // first interval
    o.lo0 = (pos && real) ? a : ((!quad && qn) ? a : -QM_INF);
    o.hi0 = (pos && real) ? b : ((neg && real) ? b : ((!quad && qp) ? a : QM_INF));
    if (pos && !real) { o.lo0 = 0.0; o.hi0 = 0.0; }
    // second interval (concave with real roots)
    o.lo1 = (neg && real) ? a : 0.0;
    o.hi1 = (neg && real) ? QM_INF : 0.0;
    return o;
}

__device__ inline Seg2 mw_feasible_intervals(double p, double q, double r, int relop, double s) {
    if (relop != RELOP_EQ) return mw_intervals_le(p, q, r - s, s - r);
    return feasible_intervals(p, q, r, relop, s);      // |f| <= s: the general rule (rare in this family)
}

struct MwBounds { double Lg, Hg; bool anyempty; };

// order-preserving map double -> u64 (the reductions over the threads of a restart are integer LDS atomics: two instructions
// per wave instead of a DPP tree per wave plus a second level through LDS)
__device__ inline unsigned long long mw_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double mw_unkey(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// workgroup barrier for LDS traffic only: the wave's LDS operations are complete (they retire in order), then s_barrier.
__device__ __attribute__((always_inline)) inline void mw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Stage B for the slack s: returns the combined bounds (identical in every thread); the gaps that cut into [Lg, Hg] are in
// W.gapa / W.gapb, their number in W.bi[3], the multiplicity of Hg in W.red[0] -- valid after the barrier this function
// ends with.  The serial thread resets the words (mw_segments) before the next barrier.
template <int SL>
__device__ __attribute__((always_inline)) inline MwBounds mw_bounds_and_gaps(const MwLds &W, int m1, int tid, int Tc, double s, const double (&t2)[SL],
                                                                              const double (&t1)[SL], const double (&t0)[SL],
                                                                              const int (&rel)[SL], DnProf &pf) {
    double L = -QM_INF, H = QM_INF;
    int mH = 0;
    bool empty = false;
    unsigned n2 = 0;
    double ga[SL], gb[SL];
#pragma unroll
    for (int j = 0; j < SL; j++) {
        const int k = 1 + tid + Tc * j;
        ga[j] = 0.0; gb[j] = 0.0;
        if (tid >= Tc || k >= m1) continue;
        if (t2[j] == 0.0 && t1[j] == 0.0) continue;   // qcqp.py:116,166
        const Seg2 iv = mw_feasible_intervals(t2[j], t1[j], t0[j], rel[j], s);
        if (iv.n == 0) { empty = true; continue; }
        const double lo = iv.lo0, hi = (iv.n == 2) ? iv.hi1 : iv.hi0;
        if (iv.n == 2) { n2 |= 1u << j; ga[j] = iv.hi0; gb[j] = iv.lo1; }
        L = lo > L ? lo : L;
        if (hi < H) { H = hi; mH = 1; } else if (hi == H) mH++;
    }
    {
        // wave level in registers (DPP), one LDS atomic per wave
        const double Lw = dn_wave_max(L), Hw = dn_wave_min(H);
        if ((tid & 63) == 0) {
            if (Lw > -QM_INF) atomicMax(W.kL, mw_key(Lw));
            if (Hw < QM_INF) atomicMin(W.kH, mw_key(Hw));
        }
    }
    if (empty) W.red[1] = 1;
    mw_barrier();
    MwBounds o;
    o.Lg = mw_unkey(*W.kL); o.Hg = mw_unkey(*W.kH); o.anyempty = W.red[1] != 0;
    // multiplicity of Hg (the serial thread adds the base interval (-inf, +inf), one more interval ending at +inf)
    if (mH > 0 && H == o.Hg) atomicAdd(&W.red[0], mH);
    pf.tick(2);
    // gaps that cut into [Lg, Hg]  (two-interval constraints only)
    if (n2) {
#pragma unroll
        for (int j = 0; j < SL; j++) {
            if (!((n2 >> j) & 1u)) continue;
            if (gb[j] > o.Lg && ga[j] <= o.Hg) {
                const int pos = atomicAdd(&W.bi[3], 1);
                if (pos < DN_GC) { W.gapa[pos] = ga[j]; W.gapb[pos] = gb[j]; }
            }
        }
    }
    mw_barrier();
    if (pf.on) pf.t[9]++;
    pf.tick(3);
    return o;
}

// serial thread: the segment list of the evaluation just made (consumes and resets the words of the reductions)
__device__ inline int mw_segments(const MwLds &W, const MwBounds &fs) {
    int ng = W.bi[3], ovf = 0;
    const int mHg = ((fs.Hg == QM_INF) ? 1 : 0) + W.red[0];
    W.bi[3] = 0; W.red[0] = 0; W.red[1] = 0;
    *W.kL = mw_key(-QM_INF); *W.kH = mw_key(QM_INF);
    if (ng > DN_GC) { ovf = 1; ng = DN_GC; }
    int ns = 0;
    if (!fs.anyempty && fs.Lg <= fs.Hg) ns = dn_sweep_segments(W.gapa, W.gapb, ng, W.seglo, W.seghi, fs.Lg, fs.Hg, mHg, &ovf);
    if (ovf) W.bi[4] = 1;
    return ns;
}

// general_minimise (cd_general.h) for a list of ONE segment [lo, hi], the usual outcome: the same decisions and values
// (utilities.py:257-288), without the two loops over the list and their re-reads
__device__ inline int mw_minimise_one(double p0, double q0, double r0, double lo, double hi, const DrawKey &dk, double *xout) {
    if (p0 == 0.0 && q0 == 0.0) {
        U4 rnd = cd_draw(dk.seed, dk.restart, dk.coord, dk.sweep_tag, dk.iter);
        (void)draw_choice(rnd, 1);
        if (__builtin_isinf(lo) || __builtin_isinf(hi)) return -1;
        *xout = draw_uniform(rnd, lo, hi);
        return 1;
    }
    const double x0 = (p0 > 0.0) ? -q0 / (2.0 * p0) : QM_NAN;
    if (lo <= x0 && x0 <= hi) { *xout = x0; return 1; }
    int err = 0, nb = 0;
    double bestf = QM_INF;
    const double fl = onevar_eval(p0, q0, r0, lo, &err), fr = onevar_eval(p0, q0, r0, hi, &err);
    if (bestf > fl) { nb = 1; bestf = fl; } else if (bestf == fl) nb++;
    if (bestf > fr) { nb = 1; bestf = fr; } else if (bestf == fr) nb++;
    if (err) return -2;
    if (nb == 0) return 0;
    int idx = 0, seen = 0;
    if (nb > 1) {
        U4 rnd = cd_draw(dk.seed, dk.restart, dk.coord, dk.sweep_tag, dk.iter);
        idx = draw_choice(rnd, nb);
    }
    if (fl == bestf) { if (seen == idx) { *xout = lo; return 1; } seen++; }
    if (fr == bestf) { if (seen == idx) { *xout = hi; return 1; } seen++; }
    return 0;
}

template <int PHASE, int SL>
__global__ __launch_bounds__(MW_TMAX) void dense_chain_mw_kernel(DenseChainArgs a) {
    extern __shared__ double smem[];
    const DenseProblem &D = a.D;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = (int)blockDim.x, nw = T >> 6;
    const int Tc = a.mw_Tc, ts = a.mw_ts;      // threads that hold constraints; the serial thread (holds the objective)
    const bool serial = tid == ts;
    const int64_t gr = blockIdx.x;
    if (gr >= a.R) return;
    if (!a.S.on[gr]) return;          // the whole workgroup leaves: no barrier is left waiting
    // the chain is the critical path of a block and shares its CUs with the products kernel of the next block (long MFMA
    // runs that leave most issue slots free): its waves go first
    __builtin_amdgcn_s_setprio(3);
    const int m1 = D.m1, m1p = D.m1p;
    DnProf pf;
    pf.start(a.prof != nullptr && gr == 0 && (serial || tid == 0));
    MwLds W;
    {
        double *sp = smem;
        W.F = sp; sp += m1p;
        W.gapa = sp; sp += DN_GC; W.gapb = sp; sp += DN_GC;
        W.seglo = sp; sp += DN_SC; W.seghi = sp; sp += DN_SC;
        W.xb = sp; sp += 16; W.dlt = sp; sp += 16;
        W.kL = (unsigned long long *)sp; W.kH = W.kL + 1; W.kV = W.kL + 2; sp += 4;
        W.red = (int *)sp; sp += 4;
        W.bx = sp; sp += 2;
        W.bi = (int *)sp;
    }
    const int b = a.b;
    const int64_t tile = gr >> 4;
    const int r = (int)(gr & 15);
    double *Xt = a.X + tile * D.n16 * 16;
    double *Ftr = a.Ft + gr * m1p;
    int rel[SL];
#pragma unroll
    for (int j = 0; j < SL; j++) {
        const int k = 1 + tid + Tc * j;
        rel[j] = (tid < Tc && k < m1) ? D.relop[k] : 0;
    }
    for (int k = tid; k < m1; k += T) W.F[k] = Ftr[k];
    if (tid < 16) { W.xb[tid] = Xt[(16 * (int64_t)b + tid) * 16 + r]; W.dlt[tid] = 0.0; }
    if (tid < 8) { W.bi[tid] = 0; W.red[tid] = 0; }
    if (tid == 0) { *W.kL = mw_key(-QM_INF); *W.kH = mw_key(QM_INF); *W.kV = mw_key(-QM_INF); }
    // per-restart state: identical in every thread
    bool live = a.S.live[gr] != 0, on = true;
    int64_t upd = a.S.upd[gr], visits = a.S.visits[gr], accepted = a.S.accepted[gr];
    int status = a.S.status[gr];
    const double slack = (PHASE == 2) ? a.slack[gr] : 0.0;
    unsigned mvmask = 0;   // coordinates of this block that moved
    mw_barrier();
    const int cmax = (D.n - 16 * (int64_t)b) < 16 ? (int)(D.n - 16 * (int64_t)b) : 16;
    const SegList SLIST{W.seglo, W.seghi, nullptr, 0};
    // function of slot j of this thread (m1: none)
    auto slot_k = [&](int j) __attribute__((always_inline)) {
        if (serial) return j == 0 ? 0 : m1;
        return tid < Tc ? 1 + tid + Tc * j : m1;
    };
    // GREG (up to two slots per thread): the thread keeps the 16 rows of G of its functions in REGISTERS -- all rows of a
    // K-split plane are requested in one batch, the planes are summed in their fixed order -- and a move is added to the rows
    // of the coordinates still to come when it is committed: the same fused multiply-adds in the same order as adding the
    // moves made so far at every visit (dense_chain_kernel), with independent loads that go out together and no per-visit
    // loop over the move list.  Beyond two slots the rows are fetched visit by visit.
    constexpr bool GREG = SL <= 2;
    typedef double mw_v16d __attribute__((ext_vector_type(16)));
    mw_v16d g[GREG ? SL : 1];        // a vector per slot: g[j][c] with the loop's c is ONE indexed register move, not a 15-way select
    if (GREG) {
#pragma unroll
        for (int j = 0; j < SL; j++) {
            const int k = slot_k(j);
#pragma unroll
            for (int cc = 0; cc < 16; cc++) g[j][cc] = 0.0;
            if (k >= m1) continue;
            const double *G0 = a.G + ((tile * 16) * 16 + r) * m1p + k;      // row cc: + cc * 16 * m1p
            for (int z = 0; z < a.zs; z += 2) {            // two planes per round trip
                double v[2][16];
                const bool two = z + 1 < a.zs;
#pragma unroll
                for (int cc = 0; cc < 16; cc++) {
                    v[0][cc] = G0[(int64_t)z * a.gz_stride + (int64_t)cc * 16 * m1p];
                    v[1][cc] = two ? G0[(int64_t)(z + 1) * a.gz_stride + (int64_t)cc * 16 * m1p] : 0.0;
                }
#pragma unroll
                for (int cc = 0; cc < 16; cc++) {
                    if (z == 0) g[j][cc] = v[0][cc]; else g[j][cc] += v[0][cc];      // K-split partials, fixed order
                    if (two) g[j][cc] += v[1][cc];
                }
            }
        }
    }
    // the diagonal entry and the linear coefficient of the coordinate to come, and the entries of the diagonal block a move
    // of the coordinate at hand would need, are requested BEFORE the barrier that waits for the serial thread (phase 2): the
    // round trips run while the segments are swept and the minimiser is found
    double nd2[GREG ? SL : 1], nql[GREG ? SL : 1], dvc[GREG ? SL : 1][15];
    auto request_next = [&](int cc) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < (GREG ? SL : 1); j++) {
            const int k = slot_k(j);
            nd2[j] = k < m1 ? a.Dg[((int64_t)cc * 16 + cc) * m1p + k] : 0.0;
            nql[j] = k < m1 ? D.qT[(16 * (int64_t)b + cc) * m1p + k] : 0.0;
        }
    };
    auto request_moves = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < (GREG ? SL : 1); j++) {
            const int k = slot_k(j);
#pragma unroll
            for (int cc = 1; cc < 16; cc++) dvc[j][cc - 1] = (k < m1 && cc > c && cc < cmax) ? a.Dg[((int64_t)cc * 16 + c) * m1p + k] : 0.0;
        }
    };
    const int c_first = a.c_lo > 0 ? a.c_lo : 0, c_end = a.c_hi < cmax ? a.c_hi : cmax;
    if (GREG) request_next(c_first < 16 ? c_first : 0);
    pf.tick(0);

    for (int c = c_first; c < c_end && on; c++) {
        const int64_t i = 16 * (int64_t)b + c;
        const double xi = W.xb[c];
        if (pf.on) pf.t[8]++;
        // ---- A. one-variable coefficients of the thread's functions (utilities.py:99-105): all requests first
        const double *Gc = a.G + ((tile * 16 + c) * 16 + r) * m1p;
        const double *Dc = a.Dg + (int64_t)(c * 16) * m1p;
        const double *qi = D.qT + i * m1p;
        double t2[SL], t1[SL], t0[SL];
        double vloc = -QM_INF;
        bool inv = false;
        if (GREG) {
            double d2[SL], ql[SL];
#pragma unroll
            for (int j = 0; j < SL; j++) { d2[j] = nd2[j]; ql[j] = nql[j]; }
#pragma unroll
            for (int j = 0; j < SL; j++) {
                const int k = slot_k(j);
                t2[j] = 0.0; t1[j] = 0.0; t0[j] = 0.0;
                if (k >= m1) continue;
                const double gs = g[j][c];                                         // c is identical in every thread
                const double u1 = 2.0 * (gs - d2[j] * xi) + ql[j];
                const double u0 = W.F[k] - xi * (d2[j] * xi + u1);
                t2[j] = d2[j]; t1[j] = u1; t0[j] = u0;
                if (PHASE == 1 && k > 0 && !(d2[j] == 0.0 && u1 == 0.0)) {
                    const double f = xi * (d2[j] * xi + u1) + u0;
                    const double v = (rel[j] == RELOP_EQ) ? fabs(f) : (f > 0.0 ? f : 0.0);
                    vloc = v > vloc ? v : vloc;
                    inv = true;
                }
            }
        } else {
            // the moves made so far in this block (identical in every thread: scalar registers)
            double dl[15];
#pragma unroll
            for (int c2 = 0; c2 < 15; c2++) dl[c2] = dn_bcast0(W.dlt[c2]);
#pragma unroll
            for (int j = 0; j < SL; j++) {
                const int k = slot_k(j);
                t2[j] = 0.0; t1[j] = 0.0; t0[j] = 0.0;
                if (k >= m1) continue;
                double gz[8], dv[15];
#pragma unroll
                for (int z = 0; z < 8; z++) gz[z] = (z < a.zs) ? Gc[(int64_t)z * a.gz_stride + k] : 0.0;
#pragma unroll
                for (int c2 = 0; c2 < 15; c2++) dv[c2] = ((mvmask >> c2) & 1u) ? Dc[(int64_t)c2 * m1p + k] : 0.0;
                const double d2 = Dc[(int64_t)c * m1p + k], ql = qi[k];
                double gs = gz[0];
#pragma unroll
                for (int z = 1; z < 8; z++) if (z < a.zs) gs += gz[z];                 // K-split partials, fixed order
#pragma unroll
                for (int c2 = 0; c2 < 15; c2++)                                       // Gauss-Seidel inside the block, in coordinate order
                    if ((mvmask >> c2) & 1u) gs = __builtin_fma(dv[c2], dl[c2], gs);
                const double u1 = 2.0 * (gs - d2 * xi) + ql;
                const double u0 = W.F[k] - xi * (d2 * xi + u1);
                t2[j] = d2; t1[j] = u1; t0[j] = u0;
                if (PHASE == 1 && k > 0 && !(d2 == 0.0 && u1 == 0.0)) {
                    const double f = xi * (d2 * xi + u1) + u0;
                    const double v = (rel[j] == RELOP_EQ) ? fabs(f) : (f > 0.0 ? f : 0.0);
                    vloc = v > vloc ? v : vloc;
                    inv = true;
                }
            }
        }
        bool moved = false;
        double xn = xi;
        visits++;
        pf.tick(1);
        if (PHASE == 2) {
            // ---- B / C. feasible set at the fixed slack, minimiser of the scalar objective
            const MwBounds fs = mw_bounds_and_gaps<SL>(W, m1, tid, Tc, slack, t2, t1, t0, rel, pf);
            if (GREG) {
                if (c + 1 < cmax) request_next(c + 1);
                request_moves(c);
            }
            if (serial) {
                SegList C = SLIST;
                C.n = mw_segments(W, fs);
                pf.tick(4);
                double xc = xi;
                DrawKey dk{a.seed, a.first_index + (uint64_t)gr, (uint32_t)i, (uint32_t)a.t | 0x80000000u, 0u};
                const int got = C.n == 1 ? mw_minimise_one(t2[0], t1[0], t0[0], W.seglo[0], W.seghi[0], dk, &xc)
                                         : general_minimise(t2[0], t1[0], t0[0], C, dk, &xc);
                const bool mv = got > 0 && fabs(xc - xi) > a.tol;
                if (mv) { W.xb[c] = xc; W.dlt[c] = xc - xi; }
                W.bx[0] = xc; W.bi[1] = got; W.bi[5] = mv ? 1 : 0;
            }
            mw_barrier();
            const int got = W.bi[1];
            xn = W.bx[0];
            if (got < 0) { status = got; live = false; on = false; }
            else if (W.bi[5]) { moved = true; upd = 0; accepted++; }
            else {
                upd++;
                if (upd == D.n) { live = false; on = false; }   // converged (qcqp.py:172-176)
            }
        } else {
            // ---- B / C. smallest achievable slack by bisection (qcqp.py:117-131)
            {
                const double vw = dn_wave_max(vloc);
                if (lane == 0 && vw > -QM_INF) atomicMax(W.kV, mw_key(vw));
            }
            if (inv) W.red[2] = 1;
            mw_barrier();
            const double viol = mw_unkey(*W.kV);
            const bool anyinv = W.red[2] != 0;
            mw_barrier();     // every thread has read the words ...
            if (serial) { *W.kV = mw_key(-QM_INF); W.red[2] = 0; }      // ... before they are reset for the next coordinate
            if (!anyinv) { status = -3; live = false; on = false; }   // ValueError (qcqp.py:117)
            else {
                double new_viol = viol, ss = -a.tol, es = viol - a.viol_tol;
                uint32_t it = 0;
                // Only the last successful step decides the point and the keyed draws are independent of each other: a
                // successful step leaves its segment list in LDS (failed steps write nothing) and the Philox draw happens
                // once, after the bisection.  A list with an unbounded piece draws at once (the reference may raise there).
                int ns_p = 0;
                uint32_t it_p = 0;
                bool pending = false;
                while (es - ss > a.tol) {
                    const double sm = (ss + es) / 2.0;
                    const MwBounds fs = mw_bounds_and_gaps<SL>(W, m1, tid, Tc, sm, t2, t1, t0, rel, pf);
                    const uint32_t itc = it++;
                    if (serial) {
                        const int ns = mw_segments(W, fs);
                        bool unb = false;
                        for (int j = 0; j < ns; j++) unb = unb || __builtin_isinf(W.seglo[j]) || __builtin_isinf(W.seghi[j]);
                        int got = 0;
                        double xc = 0.0;
                        if (ns > 0 && unb) {
                            SegList C = SLIST;
                            C.n = ns;
                            DrawKey dk{a.seed, a.first_index + (uint64_t)gr, (uint32_t)i, (uint32_t)a.t, itc};
                            got = general_minimise(0.0, 0.0, 0.0, C, dk, &xc);
                        }
                        W.bi[0] = ns; W.bi[2] = unb ? 1 : 0; W.bi[1] = got; W.bx[0] = xc;
                    }
                    mw_barrier();
                    const int ns = W.bi[0], got = W.bi[1];
                    const bool unb = W.bi[2] != 0;
                    const double xc = W.bx[0];
                    mw_barrier();     // the words are rewritten by the next evaluation
                    if (ns == 0) { ss = sm; continue; }
                    if (unb) {
                        if (got < 0) { status = got; live = false; on = false; pending = false; break; }
                        xn = xc; pending = false;
                    } else {
                        ns_p = ns; it_p = itc; pending = true;
                    }
                    new_viol = sm; es = sm;
                }
                if (pending) {
                    if (serial) {
                        double xc = 0.0;
                        SegList C = SLIST;
                        C.n = ns_p;
                        DrawKey dk{a.seed, a.first_index + (uint64_t)gr, (uint32_t)i, (uint32_t)a.t, it_p};
                        (void)general_minimise(0.0, 0.0, 0.0, C, dk, &xc);
                        W.bx[1] = xc;
                    }
                    mw_barrier();
                    xn = W.bx[1];
                }
                if (status == 0) {
                    if (new_viol < viol) { moved = true; upd = 0; accepted++; }
                    else {
                        upd++;
                        if (upd == D.n) on = false;   // "failed": leaves this sweep only (qcqp.py:138-141)
                    }
                }
                if (moved) {
                    if (serial) { W.xb[c] = xn; W.dlt[c] = xn - xi; }
                    mw_barrier();     // the move list is read by every thread at the next coordinate
                }
            }
        }
        pf.tick(5);
        // ---- D. commit: f_k(x) += delta (t2 (xn + xi) + t1)  (x_i and the move list were written by thread 0)
        if (moved) {
            const double d = xn - xi;
            mvmask |= 1u << c;
#pragma unroll
            for (int j = 0; j < SL; j++) {
                const int k = slot_k(j);
                if (k >= m1) continue;
                W.F[k] += d * (t2[j] * (xn + xi) + t1[j]);
            }
            if (GREG) {
                // Gauss-Seidel inside the block: entry (cc, c) of the diagonal block for every coordinate cc still to come
                if (PHASE == 1) request_moves(c);
#pragma unroll
                for (int j = 0; j < SL; j++) {
                    if (slot_k(j) >= m1) continue;
#pragma unroll
                    for (int cc = 1; cc < 16; cc++) if (cc > c && cc < cmax) g[j][cc] = __builtin_fma(dvc[j][cc - 1], d, g[j][cc]);
                }
            }
        }
        if (GREG && PHASE == 1 && c + 1 < cmax) request_next(c + 1);
        pf.tick(6);
    }
    mw_barrier();
    if (tid < 16) Xt[(16 * (int64_t)b + tid) * 16 + r] = W.xb[tid];
    for (int k = tid; k < m1; k += T) Ftr[k] = W.F[k];
    if (serial) {
        a.S.live[gr] = live ? 1 : 0; a.S.on[gr] = on ? 1 : 0;
        a.S.upd[gr] = upd; a.S.visits[gr] = visits; a.S.accepted[gr] = accepted;
        a.S.status[gr] = W.bi[4] ? -4 : status;
    }
    if (pf.on) {
        pf.tick(7);
        for (int q = 0; q < 10; q++) a.prof[q + (PHASE == 1 ? 0 : 16) + (serial ? 0 : 32)] += pf.t[q];
    }
}

inline size_t dense_chain_mw_lds_bytes(int m1p) { return ((size_t)m1p + MW_LDS_FIXED) * sizeof(double); }

}  // namespace qcqpmi
