// Coordinate descent phase 2 (qcqp.py:152-178), role-split pipelined kernel for the common
// Boolean / box families: every coordinate carries the same single constraint (one constraint
// class, K == 1, MAXC == 1) and the diagonal of P0 is all positive (FAST == 1, convex scalar
// objective) or all zero (FAST == 2, linear scalar objective, e.g. MAXCUT).
//
// One workgroup = 8 waves owns a tile of 16 restarts and walks the blocks of 16 coordinates:
//
//   wave 0      ("chain")  visits the 16 coordinates of block b in order, lane = restart
//                          (Gauss-Seidel: each accepted move is folded into the rest of the block
//                          through the 16x16 diagonal block of P0);
//   waves 1,2,3,5,6,7 ("mfma", two per SIMD) meanwhile compute, for the NEXT block b', the partial products
//                          G' = P0[I_b', k] X[k]  over all k outside block b on
//                          v_mfma_f64_16x16x4_f64 (K split 6 ways) -- rows of X outside block b do
//                          not change while the chain works on block b.  The A fragments do not
//                          depend on X at all: they are fetched into registers a whole iteration
//                          ahead (16-byte loads from a pair-packed copy of P0), double-buffered;
//   then wave 0 adds the 4 missing k-steps (the freshly updated rows I_b) with 4 MFMAs, sums the
//   6 partial tiles in a fixed order and starts the next chain.  Partial tiles and the staged
//   small operands are double-buffered by block parity: ONE s_barrier per block.  Wave 4 idles: it shares the
//   chain wave's SIMD.
//
// fp64 MFMA competes with fp64 VALU for a SIMD's double-precision pipe (measured: +54 % chain
// time with an MFMA wave beside it), so the chain keeps its SIMD for itself.  Any instruction
// between two fp64 MFMAs of one wave delays the next MFMA by its issue time (measured 89 instead
// of 64 cycles per MFMA with loads in the stream), hence TWO mfma waves per SIMD: one wave's loads
// and LDS reads overlap the other's matrix work.
//
// Per block the critical path is   max(fix-up + chain (16 dependent steps), mfma)  +  1 barrier;
// the MFMA work (2 n16^2 flops per restart-sweep, the roofline term) hides behind the chain.
//
// Arithmetic of the fast path: vertex xv = x_i - (G_i + q_i/2) / P_ii  (exact algebra for the
// reference's -t1/(2 t2)), projection on the interval on xv's side of the gap midpoint.  Every
// decision that is close to a tie, touches +-inf or has a vanishing objective raises `redo`: the
// block is then recomputed by the generic loop, which follows the reference's arithmetic
// literally (onevar_minimise in onevar.h).
#pragma once
#include <stdint.h>
#include "cd_phase2.h"

namespace qcqpmi {

typedef double v2d_ __attribute__((ext_vector_type(2)));

// acc += Apack[b][kk] * X rows for kk in [kk0, kk1) streaming the A fragments from L2
// (fallback for k-steps that do not fit the register prefetch, n > 16 * 3 * PFU).
template <typename XPtr>
__device__ inline v4d_ mfma_range(const double *__restrict__ Ab, XPtr Xs, int kk0, int kk1, int lane,
                                  v4d_ acc) {
    const double *ap = Ab + (int64_t)kk0 * 64 + lane;
    XPtr xp = Xs + kk0 * 64 + (lane >> 4) * 16 + (lane & 15);
    for (int k = kk0; k < kk1; k++) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[0], xp[0], acc, 0, 0, 0);
        ap += 64;
        xp += 64;
    }
    return acc;
}

constexpr int RS_NMW = 6;    // mfma waves: 1,2,3,5,6,7 (two per SIMD; wave 4 shares the chain's SIMD and idles)
constexpr int RS_PFU = 11;   // units (blocks of 16 coordinates) whose A fragments live in registers

// ------------------------------------------------------------------------------ mfma role
// Work is dealt in UNITS of 4 k-steps (= one block of 16 coordinates).  For the product of block
// `bn` the units are all blocks except `bx`, the one the chain is rewriting (bx < 0: no
// exclusion): unit j -> block  j < bx ? j : j + 1.  Wave mw owns the units [u0, u1).

// fragments of the owned units of product (bn | hole bx) -> registers (pair-packed P0 copy:
// the fragments of k-steps 2 kk2, 2 kk2 + 1 of block bn sit at ((bn KS/2 + kk2) 64 + lane) 2)
__device__ inline const v2d_ *rs_unit_ptr(const double *__restrict__ Apack2, int NB, int KS, int mw,
                                          int lane, int bn, int bx, int U) {
    const int total = (bx >= 0) ? NB - 1 : NB;
    const int u0 = total * mw / RS_NMW, u1 = total * (mw + 1) / RS_NMW;
    const int hb = (bx >= 0) ? bx : NB;
    int j = u0 + U;
    j = j < u1 ? j : u1 - 1;            // units past the owned range re-load the last one (harmless):
    j = j < 0 ? 0 : j;                  // every load is unconditional so that s_waitcnt can count them
    int bb = j < hb ? j : j + 1;
    bb = bb < NB ? bb : NB - 1;
    // wave-uniform pointer; the caller adds the lane LAST so that the loads use the
    // scalar-base + 32-bit lane offset form (no VALU address arithmetic between MFMAs: fp64 MFMA
    // and VALU share the SIMD's pipe)
    (void)lane;
    return reinterpret_cast<const v2d_ *>(Apack2) + ((int64_t)bn * (KS / 2) + 2 * bb) * 64;
}

__device__ inline void rs_prefetch(v2d_ (&ar)[2 * RS_PFU], const double *__restrict__ Apack2, int NB,
                                   int KS, int mw, int lane, int bn, int bx) {
#pragma unroll
    for (int U = 0; U < RS_PFU; U++) {
        const v2d_ *ap = rs_unit_ptr(Apack2, NB, KS, mw, lane, bn, bx, U);
        ar[2 * U] = ap[(unsigned)lane];
        ar[2 * U + 1] = ap[64u + (unsigned)lane];
    }
}

// Product of block bn (hole bx) from the fragments in `ar`; each fragment register is refilled,
// right after the MFMA that consumed it, with the fragment of the NEXT product (block bn2, hole
// bx2) -- the load lands hundreds of cycles later, long after the MFMA has read its operand.
// All X rows (B operands) of the wave are read from LDS up front: an MFMA fed by an LDS read
// issued just before it runs at ~90 cycles instead of 64 (in-order issue: the read cannot be
// issued while the wave waits for the matrix pipe).
template <typename XPtr>
__device__ inline v4d_ rs_compute(v2d_ (&ar)[2 * RS_PFU], const double *__restrict__ Apack,
                                  const double *__restrict__ Apack2, XPtr Xs, int NB, int KS, int mw,
                                  int lane, int bn, int bx, int bn2, int bx2, int dbg) {
    const int total = (bx >= 0) ? NB - 1 : NB;
    const int u0 = total * mw / RS_NMW, u1 = total * (mw + 1) / RS_NMW;
    const int hb = (bx >= 0) ? bx : NB;
    const int nu = u1 - u0;
    const int xoff = (lane >> 4) * 16 + (lane & 15);
    double bq[4 * RS_PFU];
#pragma unroll
    for (int U = 0; U < RS_PFU; U++) {
        int j = u0 + U;
        j = j < u1 ? j : u1 - 1;
        j = j < 0 ? 0 : j;
        int bb = j < hb ? j : j + 1;
        bb = bb < NB ? bb : NB - 1;
#pragma unroll
        for (int q = 0; q < 4; q++) bq[4 * U + q] = Xs[(4 * bb + q) * 64 + xoff];
    }
    v4d_ acc = {0.0, 0.0, 0.0, 0.0}, acc1 = acc, acc2 = acc, acc3 = acc;
#pragma unroll
    for (int U = 0; U < RS_PFU; U++) {
        if (U < nu) {   // wave-uniform
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[2 * U][0], bq[4 * U], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[2 * U][1], bq[4 * U + 1], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[2 * U + 1][0], bq[4 * U + 2], acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[2 * U + 1][1], bq[4 * U + 3], acc3, 0, 0, 0);
        }
        {   // unconditional: s_waitcnt can then count exactly how many loads are younger
            const v2d_ *ap = rs_unit_ptr(Apack2, NB, KS, mw, lane, bn2, bx2, U);
            ar[2 * U] = ap[(unsigned)lane];
            ar[2 * U + 1] = ap[64u + (unsigned)lane];
        }
    }
    acc = (acc + acc1) + (acc2 + acc3);
    if (nu > RS_PFU) {   // large n: the rest streams from L2
        const double *Ab = Apack + (int64_t)bn * KS * 64;
        for (int j = u0 + RS_PFU; j < u1; j++) {
            const int bb = j < hb ? j : j + 1;
            acc = mfma_range(Ab, Xs, 4 * bb, 4 * bb + 4, lane, acc);
        }
    }
    return acc;
}

// ------------------------------------------------------------------------------------ kernel

// FULL: n is a multiple of 16 (every block has 16 coordinates: no per-step bounds test in the chain)
// SYM:  the constraint is p x_i^2 + r == 0 (no linear term): the feasible set is mirrored about 0 at every slack
// PROF: in-kernel cycle counters (tools/phase_profile.py); compiled out of the production variant -- the
//       counters cost 18 VGPRs and their branches cut the block loop into basic blocks
template <bool XLDS, int FAST, bool FULL, bool SYM, bool PROF>
__global__ __launch_bounds__(512) void cd_phase2_rs_kernel(CdArgs a, const double *__restrict__ Apack,
                                                           const double *__restrict__ Apack2,
                                                           const double *__restrict__ P0,
                                                           const double *__restrict__ q0,
                                                           const double *__restrict__ rcp2d) {
    constexpr int MAXC = 1;
    extern __shared__ double smem[];
    const DevProblem &P = a.P;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for the compiler too
    const int64_t tile = blockIdx.x;
    const int64_t n16 = P.n16;
    const int NB = (int)P.NB, KS = (int)P.KS;
    double *Xg = a.X + tile * n16 * 16;
    // ---- dynamic LDS carve-up
    double *sp = smem;
    double *Xl = sp; if (XLDS) sp += n16 * 16;
    double *part2 = sp; sp += 2 * RS_NMW * 256;   // partial G tiles of the mfma waves, [c][r] layout, double-buffered
    double *Gsc = sp; sp += 256;             // complete G (+ q/2) of the current block, [c][r]
    double *Dblk2 = sp; sp += 2 * 256;       // diagonal block of P0, double-buffered by block parity
    double *hqb2 = sp; sp += 2 * 16;         // q0 / 2
    double *rtb2 = sp; sp += 2 * 16;         // 1 / P0[i,i]  (0 if P0[i,i] == 0)
    double *slk = sp; sp += 16;
    SetTable<MAXC> TC;                       // the single class: slot = restart
    TC.slots = 16;
    TC.lo = sp; sp += 2 * 16;
    TC.hi = sp; sp += 2 * 16;
    TC.n = (int *)sp; sp += 8;
    TC.slow = (int *)sp; sp += 8;
    // "every restart of the tile has converged", double-buffered by block parity: the chain wave writes the flag of interval
    // g + 1 at the end of interval g, the other waves read the flag of interval g right after barrier g -- a wave that is
    // late after the barrier can no longer see the next interval's value and leave one barrier early
    int *done = (int *)sp;

    double *Xs = XLDS ? Xl : Xg;
    if (XLDS)
        for (int64_t idx = tid; idx < n16 * 16; idx += 512) Xl[idx] = Xg[idx];
    if (tid < 16) {
        const int64_t g = tile * 16 + tid;
        slk[tid] = (g < a.R) ? a.slack[g] : 0.0;
    }
    if (tid == 0) { done[0] = 0; done[1] = 0; }
    __syncthreads();
    if (tid < 16) {
        FeasSet<MAXC> C;
        compute_set<MAXC>(P, P.krep[0], slk[tid], C);
        store_set<MAXC>(TC, tid, C);
    }
    __syncthreads();

    const int64_t gmax = a.num_iters * (int64_t)NB;

    if (wave == 4) {
        // ============================================================================ idle role
        // wave 4 lands on the chain wave's SIMD (waves w and w+4 of a workgroup share a SIMD): it
        // only keeps the barrier protocol.
        for (int64_t g = 0;; g++) {
            __syncthreads();
            if (done[g & 1] || g >= gmax) break;
        }
    } else if (wave != 0) {
        // =========================================================================== mfma role
        const int mw = wave < 4 ? wave - 1 : wave - 2;
        const int st = wave < 4 ? tid - 64 : tid - 128;   // 0..383: staging slot of this thread
        v2d_ arP[2 * RS_PFU];
        long long qc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = 0;
#define QTICK(slot) if (PROF && a.prof && wave == 1) { long long now_ = (long long)__builtin_amdgcn_s_memtime(); qc[slot] += now_ - tq; tq = now_; }
        // staging of the small operands of a block: diagonal block of P0 (256 entries over 192
        // threads) and q/2, 1/P_ii of its 16 coordinates (threads 0..15).  Loads are issued early,
        // the LDS stores come after the matrix work.
        double st_d0 = 0.0, st_q = 0.0, st_r = 0.0;
        auto stage_load = [&](int bn) {
            if (st < 256) st_d0 = P0[(16 * (int64_t)bn + (st >> 4)) * n16 + 16 * bn + (st & 15)];
            if (st < 16) { st_q = q0[16 * (int64_t)bn + st]; st_r = rcp2d[16 * (int64_t)bn + st]; }
        };
        auto stage_store = [&](int buf) {
            double *Dblk = Dblk2 + buf * 256;
            if (st < 256) Dblk[st] = st_d0;
            if (st < 16) { hqb2[buf * 16 + st] = 0.5 * st_q; rtb2[buf * 16 + st] = st_r + st_r; }
        };
        auto store_part = [&](const v4d_ &acc, int buf) {
            double *part = part2 + buf * RS_NMW * 256;
#pragma unroll
            for (int v = 0; v < 4; v++) part[mw * 256 + ((lane >> 4) + 4 * v) * 16 + (lane & 15)] = acc[v];
        };
        // prologue: full G of block 0 (no exclusion), then the fragments of iteration 0's product
        rs_prefetch(arP, Apack2, NB, KS, mw, lane, 0, -1);
        stage_load(0);
        {
            v4d_ acc = rs_compute(arP, Apack, Apack2, Xs, NB, KS, mw, lane, 0, -1, 1 % NB, 0, 0);
            store_part(acc, 0);
            stage_store(0);
        }
        if (PROF && a.prof) tq = (long long)__builtin_amdgcn_s_memtime();
        // iteration g computes the product for block b(g+1) with hole b(g) from set S_, and
        // prefetches the fragments of iteration g+1 (block b(g+2), hole b(g+1)) into set T_.
#define RS_MFMA_ITER(S_, T_)                                                          \
        {                                                                             \
            const int b = (int)(g % NB);                                              \
            QTICK(0)                                                                  \
            __syncthreads();                                                          \
            if (done[g & 1] || g >= gmax) break;                                            \
            stage_load((b + 1) % NB);                                                 \
            QTICK(1)                                                                  \
            QTICK(2)                                                                  \
            v4d_ acc = {0.0, 0.0, 0.0, 0.0};                                          \
            if (!(a.dbg & 1)) acc = rs_compute(arP, Apack, Apack2, Xs, NB, KS, mw, lane, (b + 1) % NB, b, (b + 2) % NB, (b + 1) % NB, a.dbg); \
            QTICK(3)                                                                  \
            store_part(acc, (int)((g + 1) & 1));                                      \
            stage_store((int)((g + 1) & 1));                                          \
            g++;                                                                      \
        }
        for (int64_t g = 0;;) {
            RS_MFMA_ITER(arP, arP)
        }
#undef RS_MFMA_ITER
        if (PROF && a.prof && tid == 64)
            for (int k = 0; k < 8; k++) a.prof[tile * 16 + 8 + k] = qc[k];
#undef QTICK
    } else {
        // ========================================================================== chain role
        __builtin_amdgcn_s_setprio(3);
        const int r = lane & 15;
        const int64_t gr = tile * 16 + r;
        // feasible set of this lane's restart: registers for the whole kernel
        StepTab U;
        U.l0 = TC.lo[r]; U.h0 = TC.hi[r]; U.l1 = TC.lo[16 + r]; U.h1 = TC.hi[16 + r];
        U.n = TC.n[r]; U.slow = TC.slow[r];
        {
            const bool two = U.n >= 2;
            U.mid = two ? 0.5 * (U.h0 + U.l1) : QM_INF;
            U.thr = two ? 1e-7 * (U.l1 - U.h0) : 0.0;
        }
        ChainState S;
        S.fcur = 0.0; S.upd_counter = 0; S.visits = 0; S.accepted = 0; S.sweeps = 0;
        S.conv = true; S.status = 0;
        // ALL 64 lanes walk the chain: lanes l, l+16, l+32, l+48 carry the same restart (r = l & 15) and
        // compute the same values.  No divergent region = fix-up and chain are one basic block, and the
        // scheduler starts the chain's LDS reads under the fix-up MFMAs.
        if (gr < a.R) {
            S.conv = a.flag[gr] ? false : true;
            S.fcur = a.f0cur[gr];
        }
        double afix[4] = {0.0, 0.0, 0.0, 0.0};   // A fragments of the next fix-up
        long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = 0;
#define PROF_TICK(slot) if (PROF && a.prof) { long long now_ = (long long)__builtin_amdgcn_s_memtime(); pc[slot] += now_ - tp; tp = now_; }
        if (PROF && a.prof) tp = (long long)__builtin_amdgcn_s_memtime();
        for (int64_t g = 0;; g++) {
            const int b = (int)(g % NB);
            const int64_t t = g / NB;
            const int bprev = (g > 0) ? (int)((g - 1) % NB) : -1;
            const int cur = (int)(g & 1);
            const double *Dblk = Dblk2 + cur * 256, *hqb = hqb2 + cur * 16, *rtb = rtb2 + cur * 16;
            const double *part = part2 + cur * RS_NMW * 256;
            PROF_TICK(0)
            __syncthreads();                  // (1) part/Dblk/hq/rt of block b and X rows of bprev are ready
            PROF_TICK(1)
            if (done[g & 1] || g >= gmax) break;
            double xb[16], gb[16];
            {
                // ---- fix-up: the 4 k-steps of the block the chain has just updated (A fragments
                // were fetched during that chain), then the partials, in a fixed order
                v4d_ acc = {0.0, 0.0, 0.0, 0.0};
                if (bprev >= 0) {
                    const int xo = (4 * bprev) * 64 + (lane >> 4) * 16 + (lane & 15);
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afix[u], Xs[xo + u * 64], acc, 0, 0, 0);
                }
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int c = (lane >> 4) + 4 * v;
                    double s = acc[v];
#pragma unroll
                    for (int w = 0; w < RS_NMW; w++) s += part[w * 256 + c * 16 + (lane & 15)];
                    Gsc[c * 16 + (lane & 15)] = s + hqb[c];
                }
                // same wave: LDS operations complete in order, no barrier needed before reading Gsc
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    gb[c] = Gsc[c * 16 + r];
                    xb[c] = Xs[(16 * b + c) * 16 + r];
                }
                if (b == 0 && !S.conv) S.sweeps++;
            }
            PROF_TICK(2)
            PROF_TICK(3)
            {   // A fragments for the fix-up of the NEXT block: k-steps of this block
                const int bn = (b + 1) % NB;
                const double *ap = Apack + ((int64_t)bn * KS + 4 * b) * 64 + lane;
#pragma unroll
                for (int u = 0; u < 4; u++) afix[u] = ap[u * 64];
            }
            {
                const int cmax = (P.n - 16 * (int64_t)b) < 16 ? (int)(P.n - 16 * (int64_t)b) : 16;  // uniform
                const bool act = !S.conv;
                const bool actn = act && U.n > 0;
                // SYM: [-b, -a] u [a, b]  (one interval [-b, b]: a = 0)
                const double syma = (U.n >= 2) ? U.l1 : 0.0, symb = (U.n >= 2) ? U.h1 : U.h0;
                int upd = (int)S.upd_counter, accn = 0;
                unsigned mv = 0;     // FULL: bit 15 - c = coordinate c moved (instead of two counters per step)
                bool allfar = true;  // no decision close to a tie, no NaN
                double fcur = S.fcur;
                // row c of the diagonal block (wave-uniform LDS broadcast reads), fetched one step ahead
                // into ping-pong register sets (no copies).  The LDS addresses are laundered through
                // VGPRs so that every read is `ds_read base, offset:imm` (< 2 KB) with no address math.
                typedef __attribute__((address_space(3))) const double lds_cdouble;
                unsigned dva = (unsigned)(uintptr_t)(lds_cdouble *)Dblk, rta = (unsigned)(uintptr_t)(lds_cdouble *)rtb;
                asm volatile("v_mov_b32 %0, %1" : "=v"(dva) : "v"(dva));
                asm volatile("v_mov_b32 %0, %1" : "=v"(rta) : "v"(rta));
                lds_cdouble *Dv = (lds_cdouble *)(uintptr_t)dva, *rtv = (lds_cdouble *)(uintptr_t)rta;
                double dr[2][16], t2v[2], rv[2];
                t2v[0] = Dv[0];
                rv[0] = rtv[0];
#pragma unroll
                for (int c2 = 1; c2 < 16; c2++) dr[0][c2] = Dv[c2];
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    if (!FULL && c >= cmax) continue;   // wave-uniform
                    asm volatile("" ::: "memory");     // LDS reads stay exactly one step ahead (register pressure)
                    if (c + 1 < 16) {
                        t2v[(c + 1) & 1] = Dv[(c + 1) * 16 + (c + 1)];
                        rv[(c + 1) & 1] = rtv[c + 1];
#pragma unroll
                        for (int c2 = c + 2; c2 < 16; c2++) dr[(c + 1) & 1][c2] = Dv[(c + 1) * 16 + c2];
                    }
                    const double t2 = t2v[c & 1], rt = rv[c & 1];
                    const double xi = xb[c];
                    const double g2 = gb[c];                      // G_i + q_i / 2  (includes P_ii x_i)
                    double pick;
                    bool far;
                    if (FAST == 1 && SYM) {
                        // sets mirrored about 0 (x_i^2 within a band): clamp |xv|, put the sign back
                        const double xv = __builtin_fma(-g2, rt, xi);          // vertex of the scalar objective
                        pick = __builtin_copysign(fmin(fmax(fabs(xv), syma), symb), xv);
                        far = fabs(xv) > U.thr;                                // false for NaN as well
                    } else if (FAST == 1) {
                        const double xv = __builtin_fma(-g2, rt, xi);
                        const double p0 = fmin(fmax(xv, U.l0), U.h0);
                        const double p1 = fmin(fmax(xv, U.l1), U.h1);
                        pick = (xv > U.mid) ? p1 : p0;
                        far = fabs(xv - U.mid) > U.thr;
                    } else {
                        // linear scalar objective: slope t1 = 2 (G_i + q_i/2), extreme end point against it
                        const double L = U.l0;
                        const double H = (U.n >= 2) ? U.h1 : U.h0;
                        pick = (g2 > 0.0) ? L : H;
                        far = 2.0 * fabs(g2) * (fabs(L) + fabs(H)) > 1e-9 * fabs(fcur) + 1e-300 && pick == pick;
                    }
                    allfar = allfar && far;
                    const double dlt = pick - xi;
                    const bool moved = actn && fabs(dlt) > a.tol;
                    const double delta = moved ? dlt : 0.0;
                    xb[c] = moved ? pick : xi;                    // in place: committed to LDS only if no redo
                    // f(x + delta e_i) - f(x) = delta (2 (P x)_i + q_i + P_ii delta) = delta (t2 delta + 2 g2):
                    // g2 = G_i + q_i / 2 already contains P_ii x_i
                    fcur = __builtin_fma(delta, __builtin_fma(t2, delta, g2 + g2), fcur);
                    if (FULL) {
                        // pinned in place (a plain expression is sunk to the end of the block by the
                        // scheduler, keeping 16 compare masks alive = SGPR spills)
                        const unsigned b01 = moved ? 1u : 0u;
                        asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(mv) : "v"(b01));
                    } else {
                        accn += moved ? 1 : 0;
                        upd = moved ? 0 : upd + 1;
                    }
#pragma unroll
                    for (int c2 = c + 1; c2 < 16; c2++) gb[c2] = __builtin_fma(dr[c & 1][c2], delta, gb[c2]);
                }
                if (FULL) {
                    accn = __builtin_popcount(mv);
                    upd = mv ? __builtin_ctz(mv) : upd + 16;
                }
                bool redo = !allfar;
                redo = act && U.n > 0 && (redo || U.slow != 0);
                if (__builtin_amdgcn_ballot_w64(redo) == 0ull) {
                    if (act) {
                        // n consecutive rejections = convergence (qcqp.py:172-176).  Visits past that
                        // point inside this block changed nothing (deterministic map) and are not counted.
                        S.fcur = fcur; S.accepted += accn;
                        const int over = upd - (int)P.n;
                        S.visits += cmax - (over > 0 ? over : 0);
                        S.upd_counter = upd;
                        if (over >= 0) S.conv = true;
                    }
                    if (lane < 16) {
#pragma unroll
                        for (int c = 0; c < 16; c++) Xs[(16 * b + c) * 16 + r] = xb[c];
                    }
                } else {
                    // ---- generic loop (rare): the reference's arithmetic, state in LDS (Gsc, X rows)
                    if (PROF) pc[6]++;
                    for (int c = 0; c < cmax; c++) {
                        const int64_t i = 16 * (int64_t)b + c;
                        FeasSet<MAXC> C;
                        // (rare path: the set comes back from the LDS table, not from registers held all kernel long)
                        C.n = U.n; C.lo[0] = TC.lo[r]; C.hi[0] = TC.hi[r]; C.lo[1] = TC.lo[16 + r]; C.hi[1] = TC.hi[16 + r];
                        const double t2g = Dblk[c * 16 + c];
                        const double xi = Xs[i * 16 + r];
                        const double hq = hqb[c];
                        const double t1 = 2.0 * ((Gsc[c * 16 + r] - hq) - t2g * xi) + (hq + hq);
                        const double t0 = S.fcur - xi * (t2g * xi + t1);
                        DrawKey dk{a.seed, a.first_index + (uint64_t)gr, (uint32_t)i, (uint32_t)t | 0x80000000u, 0u};
                        double xn = xi;
                        int got = S.conv ? 0 : onevar_minimise<MAXC>(t2g, t1, t0, C, dk, &xn);
                        bool moved;
                        double delta;
                        chain_commit<MAXC>(S, got, xn, xi, t2g, t1, t0, a.tol, P.n, moved, delta);
                        if (moved) {
                            Xs[i * 16 + r] = xn;
                            for (int c2 = c + 1; c2 < 16; c2++) Gsc[c2 * 16 + r] += Dblk[c * 16 + c2] * delta;
                        }
                    }
                }
            }
            const unsigned long long live = __builtin_amdgcn_ballot_w64(lane < 16 && !S.conv);
            if (lane == 0) done[(g + 1) & 1] = (live == 0ull) ? 1 : 0;
            if (PROF) pc[5]++;
        }
        if (lane < 16 && gr < a.R) {
            a.visits[gr] = S.visits; a.accepted[gr] = S.accepted; a.sweeps[gr] = S.sweeps;
            a.status[gr] = S.status;
            if (a.f0out && a.flag[gr]) a.f0out[gr] = S.fcur;   // tracked exactly through every accepted move
        }
        if (PROF && a.prof && tid == 0)
            for (int k = 0; k < 8; k++) a.prof[tile * 16 + k] = pc[k];
#undef PROF_TICK
    }
    __syncthreads();
    if (a.mvout) {
        // max violation of the final points, same expression as eval_kernel: (p x + q) x + r of the one
        // constraint every coordinate carries (single class, one constraint per coordinate)
        const int e0 = P.cptr[P.krep[0]];
        const double cp = P.cp[e0], cq = P.cq[e0], cr = P.cr[e0];
        const int rel = P.crel[e0];
        const int r = tid & 15, slot = tid >> 4;
        double v = -QM_INF;
        for (int64_t i = slot; i < P.n; i += 32) {
            const double x = Xs[i * 16 + r];
            const double f = (cp * x + cq) * x + cr;
            const double w = (rel == RELOP_EQ) ? fabs(f) : (f > 0.0 ? f : 0.0);
            v = w > v ? w : v;
        }
        double *red = part2;                 // 512 doubles of the partial-tile area, free by now
        red[tid] = v;
        __syncthreads();
        if (tid < 16) {
            const int64_t g = tile * 16 + tid;
            double m = -QM_INF;
            for (int s2 = 0; s2 < 32; s2++) { const double w = red[s2 * 16 + tid]; m = w > m ? w : m; }
            if (g < a.R && a.flag[g]) a.mvout[g] = m;
        }
    }
    if (XLDS)
        for (int64_t idx = tid; idx < n16 * 16; idx += 512) Xg[idx] = Xl[idx];
}

}  // namespace qcqpmi
