// Coordinate descent phase 2 (qcqp.py:152-178): role-split pipelined kernel, second generation, for the
// family the headline benchmark runs -- every coordinate carries the one constraint  p x_i^2 + r == 0
// (single class, feasible sets mirrored about 0), every P0[i,i] > 0, n a multiple of 16 and the tile of
// 16 restarts resident in LDS (n <= 1024).  Everything else stays on cd_phase2_rs.h / cd_phase2.h.
//
// Same blocked Gauss-Seidel and the same decisions as cd_phase2_rs.h (vertex of the scalar objective, clamp
// onto the mirrored band, |move| > tol, near-tie detection with replay in the reference's arithmetic); what
// changed is how the work is laid out:
//
//   chain wave (wave 0, alone on its SIMD)
//     * QUAD layout: lane 4 r + g carries restart r and OWNS the columns c = 4 v + g (v = 0..3) of the
//       block: at step c the owner quad-lane decides, the move travels to the other three lanes of the
//       quad with two v_mov_b32 quad_perm DPPs, and every lane folds it into the <= 4 columns it owns
//       (the first generation carried each restart 4x redundantly: 120 fmas and ~75 LDS broadcast reads
//       per block and lane; here 40 fmas and 24 reads).  The matrix-core accumulator layout
//       (row = (l >> 4) + 4 v) is the same column assignment, so the partial tiles are stored [v][4 r + g]
//       and every lane sums exactly the 4 entries it needs: no G tile round trip through LDS.
//     * bookkeeping that does not feed the next step -- objective tracking, near-tie test, move mask --
//       is computed once per block from the values the lane still holds (G of a finished column is never
//       updated again: the masked diagonal block has zeros there).
//     * the 4 k-steps of the block just rewritten (the "fix-up") run straight after the commit of the block, together
//       with the chain wave's own share of the next product (the last CS blocks of the contraction).
//   mfma waves (1, 2, 3, 5, 6, 7; two per SIMD on SIMDs 1-3)
//     * block j of 16 coordinates belongs to SIMD j % 3 (the chain keeps the last CS blocks); the two waves of a SIMD
//       take ALTERNATE products (product i = block row b(i), read by the chain at the start of interval i), each over
//       all blocks of its SIMD.  B operands (those X rows) stay in registers for the whole kernel; a product re-reads
//       only the blocks committed since the wave's previous product.
//     * A fragments stream from the pair-packed copy of P0 through a ring of RQ_PFU units with buffer loads (descriptor +
//       scalar offset per unit + one lane-offset register): a unit is 4 MFMAs, 2 loads and 4 scalar instructions.
//     * TWO holes: the product for block row b(i) leaves out the block being rewritten b(i-1) AND the one rewritten just
//       before b(i-2); the chain supplies both -- after committing block b it multiplies it with the fragments of the next
//       two rows (8 MFMAs) and carries the second tile in registers for one block.  A product may start as soon as
//       interval i - 3 is committed and has almost three intervals to finish.
//   staging wave (wave 4, on the chain's SIMD): fetches the small operands of the next block (masked diagonal block,
//     diagonal, q/2, 1/P_ii) one block ahead.
//   synchronisation
//     * NO s_barrier inside the block loop.  Producer / consumer words in LDS (RQ_CONS, RQ_COMMIT, RQ_STOP, one
//       progress word per producer): the chain waits for the three partial tiles of the product's parity and the staged
//       operands of its block, a producer for the commit it depends on and for the release of the slot it is about to
//       overwrite.  Partial tiles and staged operands are double-buffered by parity.
//   Measurements, the variants that did not pay (three holes, rings of 7 / 11 units, priorities, a stream token, the
//   chain's share on wave 4) and what was learnt about the matrix pipe: DESIGN.md section 4.1.
#pragma once
#include <stdint.h>
#include "cd_phase2_rs.h"

namespace qcqpmi {

constexpr int RQ_NMW = 6;     // mfma waves: two per SIMD on three SIMDs, taking turns
constexpr int RQ_NSIMD = 3;   // SIMDs that multiply (the fourth runs the chain)
constexpr int RQ_PFU = 5;     // A-fragment ring of an mfma wave: units (blocks of 16 coordinates) resident at a time
constexpr int RQ_RND = 4;     // passes over the ring per product
constexpr int RQ_PERS = 20;   // units whose B operands stay in registers; any others are re-read for every product
constexpr int RQ_MAXU = 20;   // blocks one SIMD can own (<= RQ_RND * RQ_PFU): n = 1024 needs the chain to take >= 4 blocks.
                              // (22 units = 5 passes + 2 re-read units measured 2.5 % slower at the same split, and smaller
                              //  chain shares do not pay: the mfma waves become the bottleneck)
constexpr int RQ_CSMAX = 6;   // blocks the chain wave can own

// LDS doubles besides the X tile
constexpr int RQ_LDS_COMMON = 2 * RQ_NSIMD * 256 + 256 + 2 * 256 + 2 * 16 + 2 * 16 + 2 * 16 + 16 + 4 * 16 + 8 + 8 + 8;

// the product loop's look-ahead loads reach unit RQ_MAXU - 1 of the X tile (at the start of the allocation) whatever n is
constexpr size_t RQ_LDS_MIN = (size_t)((RQ_NSIMD - 1) * 256 + (12 * (RQ_MAXU - 1) + 3) * 64 + 64) * 8;

template <int CTRL>
__device__ __attribute__((always_inline)) inline double rq_quad_bcast(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __attribute__((always_inline)) inline double rq_quad_sum(double v) {
    const double a = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), 0xB1, 0xf, 0xf, true),
                                      __builtin_amdgcn_mov_dpp(__double2loint(v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    const double s = v + a;
    const double b = __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(s), 0x4E, 0xf, 0xf, true),
                                      __builtin_amdgcn_mov_dpp(__double2loint(s), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    return s + b;
}

__device__ __attribute__((always_inline)) inline unsigned rq_quad_or(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, true);
    v |= (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, true);
    return v;
}

// Ownership of the contraction: the chain wave multiplies the last CS blocks of 16 coordinates itself, the others are
// dealt cyclically to the three multiplying SIMDs: block j < NB - CS belongs to SIMD j % 3, unit j / 3.
struct RqOwn {
    int first;    // first owned block
    int stride;   // distance between owned blocks
    int nu;       // number of owned blocks
    int NB;
};

__device__ __attribute__((always_inline)) inline RqOwn rq_own(int NB, int CS, int mw) {
    RqOwn o;
    const int cs = CS < NB ? CS : 0;
    const int rest = NB - cs;
    o.NB = NB;
    if (mw >= RQ_NSIMD) { o.first = rest; o.stride = 1; o.nu = cs; }
    else { o.first = mw; o.stride = RQ_NSIMD; o.nu = mw < rest ? (rest - mw + RQ_NSIMD - 1) / RQ_NSIMD : 0; }
    return o;
}

// block held by register slot U (clamped to a valid block: slots past the owned ones are loaded, never multiplied)
__device__ __attribute__((always_inline)) inline int rq_block(const RqOwn &o, int U) {
    const int bb = o.first + U * o.stride;
    return bb < o.NB ? bb : o.NB - 1;
}

// slot of block j, or -1 if the wave does not own it
__device__ __attribute__((always_inline)) inline int rq_slot(const RqOwn &o, int j) {
    const int d = j - o.first;
    if (d < 0 || d % o.stride != 0) return -1;
    const int U = d / o.stride;
    return U < o.nu ? U : -1;
}

// A fragments (pair-packed copy: k-steps 2 kk2, 2 kk2 + 1 of block row bn at ((bn KS/2 + kk2) 64 + lane) 2) of the
// owned blocks for the product of block row bn -> registers
template <int NU>
__device__ __attribute__((always_inline)) inline void rq_load_A(v2d_ (&ar)[2 * NU], const double *__restrict__ Apack2, int KS, const RqOwn &o, int lane, int bn) {
#pragma unroll
    for (int U = 0; U < NU; U++) {
        const v2d_ *ap = reinterpret_cast<const v2d_ *>(Apack2) + ((int64_t)bn * (KS / 2) + 2 * rq_block(o, U)) * 64;
        ar[2 * U] = ap[(unsigned)lane];
        ar[2 * U + 1] = ap[64u + (unsigned)lane];
    }
}

// product of block row bn over the owned blocks except slots `hs`, `hs2` (the block the chain is rewriting and the one it
// rewrote just before: the chain supplies both itself; -1 = none); every
// fragment register is refilled right after the MFMAs that consumed it with the fragment of block row bn2
template <int NU>
__device__ __attribute__((always_inline)) inline v4d_ rq_product(v2d_ (&ar)[2 * NU], const double (&bq)[4 * NU], const double *__restrict__ Apack2, int KS,
                                  const RqOwn &o, int lane, int hs, int hs2, int bn2, v4d_ acc0) {
    v4d_ acc = acc0, acc1 = {0.0, 0.0, 0.0, 0.0}, acc2 = acc1, acc3 = acc1;
#pragma unroll
    for (int U = 0; U < NU; U++) {
        if (U < o.nu && U != hs && U != hs2) {   // wave-uniform
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[2 * U][0], bq[4 * U], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[2 * U][1], bq[4 * U + 1], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[2 * U + 1][0], bq[4 * U + 2], acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[2 * U + 1][1], bq[4 * U + 3], acc3, 0, 0, 0);
        }
        {   // unconditional refill: s_waitcnt can count the loads
            const v2d_ *ap = reinterpret_cast<const v2d_ *>(Apack2) + ((int64_t)bn2 * (KS / 2) + 2 * rq_block(o, U)) * 64;
            ar[2 * U] = ap[(unsigned)lane];
            ar[2 * U + 1] = ap[64u + (unsigned)lane];
        }
    }
    return (acc + acc1) + (acc2 + acc3);
}

// B operands of the owned blocks from the X tile in LDS (MFMA B layout: lane l <- X[4 kk + (l >> 4)][l & 15])
template <int NU>
__device__ __attribute__((always_inline)) inline void rq_load_B(double (&bq)[4 * NU], const double *Xs, const RqOwn &o, int lane) {
    const int xoff = (lane >> 4) * 16 + (lane & 15);
#pragma unroll
    for (int U = 0; U < NU; U++) {
        const int bb = rq_block(o, U);
#pragma unroll
        for (int q = 0; q < 4; q++) bq[4 * U + q] = Xs[(4 * bb + q) * 64 + xoff];
    }
}

// the block in slot `us` has been rewritten by the chain: refresh those 4 operands
template <int NU>
__device__ __attribute__((always_inline)) inline void rq_refresh_B(double (&bq)[4 * NU], const double *Xs, const RqOwn &o, int lane, int us) {
    const int xoff = (lane >> 4) * 16 + (lane & 15);
    const int bb = rq_block(o, us < 0 ? 0 : us);
    const double n0 = Xs[(4 * bb + 0) * 64 + xoff], n1 = Xs[(4 * bb + 1) * 64 + xoff];
    const double n2 = Xs[(4 * bb + 2) * 64 + xoff], n3 = Xs[(4 * bb + 3) * 64 + xoff];
#pragma unroll
    for (int U = 0; U < NU; U++) {
        const bool hit = (U == us);      // wave-uniform
        bq[4 * U + 0] = hit ? n0 : bq[4 * U + 0];
        bq[4 * U + 1] = hit ? n1 : bq[4 * U + 1];
        bq[4 * U + 2] = hit ? n2 : bq[4 * U + 2];
        bq[4 * U + 3] = hit ? n3 : bq[4 * U + 3];
    }
}

// ---- synchronisation words in LDS (no s_barrier inside the block loop: waves only wait for what they consume)
//   [0] cons    = g + 1 once the chain has read the partial tiles of interval g
//   [1] commit  = g + 1 once the chain has committed the block of interval g to the X tile (and is done with its staged operands)
//   [2] stop    != 0: leave the loop
//   [4 + w]     iterations published by producer w: the six mfma waves (partial tiles), w = 6: the staging wave (small
//               operands of the block); one word PER WAVE: a shared counter would let a wave that runs ahead stand in for
//               one that lags
// LDS operations of one wave complete in program order, so "data, then flag" needs no wait on the producer side and
// "flag, then data" none on the consumer side.
enum { RQ_CONS = 0, RQ_COMMIT = 1, RQ_STOP = 2, RQ_PARTS = 4 };

typedef int rq_i4 __attribute__((ext_vector_type(4)));
// explicit LDS address space: a volatile access through a generic pointer becomes a FLAT instruction (vmcnt + lgkmcnt,
// not ordered with the wave's DS queue) -- the protocol needs plain ds_read / ds_write
typedef __attribute__((address_space(3))) int rq_lds_int;
typedef __attribute__((address_space(3))) rq_i4 rq_lds_i4;

__device__ __attribute__((always_inline)) inline rq_i4 rq_sync_read(rq_lds_int *sy) {
    rq_i4 v = *(volatile rq_lds_i4 *)sy;                     // one ds_read_b128
    asm volatile("" ::: "memory");                           // nothing that follows may be read before the flags
    v[0] = __builtin_amdgcn_readfirstlane(v[0]); v[1] = __builtin_amdgcn_readfirstlane(v[1]);
    v[2] = __builtin_amdgcn_readfirstlane(v[2]); v[3] = __builtin_amdgcn_readfirstlane(v[3]);
    return v;
}

__device__ __attribute__((always_inline)) inline void rq_sync_write(rq_lds_int *sy, int which, int value, int lane) {
    asm volatile("" ::: "memory");                           // data first, then the flag (DS queue is in order per wave)
    if (lane == 0) *(volatile rq_lds_int *)(sy + which) = value;
    asm volatile("" ::: "memory");
}

// CS: blocks of the contraction the chain wave multiplies itself (0..RQ_CSMAX)
template <int CS, bool PROF>
__global__ __launch_bounds__(512) void cd_phase2_q_kernel(CdArgs a, const double *__restrict__ Apack,
                                                          const double *__restrict__ Apack2,
                                                          const double *__restrict__ P0,
                                                          const double *__restrict__ q0,
                                                          const double *__restrict__ rcp2d) {
    constexpr int MAXC = 1;
    constexpr int CSU = CS > 0 ? CS : 1;
    extern __shared__ double smem[];
#define RQ_TRACE(idx, e) if (PROF && a.prof && blockIdx.x == 0 && (idx) < 32 && (threadIdx.x & 63) == 0)                       \
        a.prof[(int64_t)gridDim.x * 16 + (threadIdx.x >> 6) * 256 + 8 * (idx) + (e)] = (long long)__builtin_amdgcn_s_memtime();
    const DevProblem &P = a.P;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tile = blockIdx.x;
    const int64_t n16 = P.n16;
    const int NB = (int)P.NB, KS = (int)P.KS;
    double *Xg = a.X + tile * n16 * 16;
    // ---- dynamic LDS carve-up
    double *sp = smem;
    double *Xs = sp; sp += n16 * 16;
    double *part2 = sp; sp += 2 * RQ_NSIMD * 256; // partial G tiles (one per multiplying SIMD and product), [v][4 r + g], by product parity
    double *fixp = sp; sp += 256;                  // the chain wave's own plane (fix-up + its share); the generic path's G tile
    double *DU2 = sp; sp += 2 * 256;               // strictly upper triangle of the diagonal block (zeros elsewhere), by parity
    double *dg2 = sp; sp += 2 * 16;                // P0[i,i]
    double *hqb2 = sp; sp += 2 * 16;               // q0 / 2
    double *rtb2 = sp; sp += 2 * 16;               // 1 / P0[i,i]
    double *slk = sp; sp += 16;
    SetTable<MAXC> TC;
    TC.slots = 16;
    TC.lo = sp; sp += 2 * 16;
    TC.hi = sp; sp += 2 * 16;
    TC.n = (int *)sp; sp += 8;
    TC.slow = (int *)sp; sp += 8;
    rq_lds_int *sy = (rq_lds_int *)(int *)sp;      // synchronisation words (16-byte aligned: sp advances in doubles from a 16-byte base)

    for (int64_t idx = tid; idx < n16 * 16; idx += 512) Xs[idx] = Xg[idx];
    if (tid < 16) {
        const int64_t g = tile * 16 + tid;
        slk[tid] = (g < a.R) ? a.slack[g] : 0.0;
    }
    if (tid < 16) *(volatile rq_lds_int *)(sy + tid) = (tid >= RQ_PARTS && tid < RQ_PARTS + 3) ? -1 : 0;   // words of the even-product waves start odd
    __syncthreads();
    if (tid < 16) {
        FeasSet<MAXC> C;
        compute_set<MAXC>(P, P.krep[0], slk[tid], C);
        store_set<MAXC>(TC, tid, C);
    }
    __syncthreads();

    const int64_t gmax = a.num_iters * (int64_t)NB;

    if (wave == 4) {
        // ========================================================================= staging role
        // wave 4 shares the chain wave's SIMD (no matrix work there while the chain runs); it fetches the small operands
        // of the next block -- strictly upper triangle of the 16 x 16 diagonal block of P0 (zeros elsewhere), diagonal,
        // q/2, 1/P_ii -- one block ahead and drops them into the slot the chain has just released.
        double d4[4], sq = 0.0, sr = 0.0, sd = 0.0;
        auto stage_load = [&](int bn) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int st = lane + 64 * e;
                d4[e] = P0[(16 * (int64_t)bn + (st >> 4)) * n16 + 16 * bn + (st & 15)];
            }
            if (lane < 16) { sq = q0[16 * (int64_t)bn + lane]; sr = rcp2d[16 * (int64_t)bn + lane];
                             sd = P0[(16 * (int64_t)bn + lane) * n16 + 16 * bn + lane]; }
        };
        auto stage_store = [&](int buf) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int st = lane + 64 * e;
                DU2[buf * 256 + st] = ((st & 15) > (st >> 4)) ? d4[e] : 0.0;
            }
            if (lane < 16) { hqb2[buf * 16 + lane] = 0.5 * sq; rtb2[buf * 16 + lane] = sr + sr; dg2[buf * 16 + lane] = sd; }
        };
        int published = 0;
        stage_load(0);
        stage_store(0);
        rq_sync_write(sy, RQ_PARTS + RQ_NMW, ++published, lane);
        int b = 0;
        for (int64_t g = 0; g < gmax; g++) {
            const int bn = (b + 1 == NB) ? 0 : b + 1;
            stage_load(bn);
            bool stop = false;
            for (;;) {     // slot (g + 1) & 1 was in use during interval g - 1
                const rq_i4 s4 = rq_sync_read(sy);
                if (s4[RQ_STOP]) { stop = true; break; }
                if (s4[RQ_COMMIT] >= (int)g) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (stop) break;
            stage_store((int)((g + 1) & 1));
            rq_sync_write(sy, RQ_PARTS + RQ_NMW, ++published, lane);
            RQ_TRACE(g, 0)
            b = bn;
        }
    } else if (wave != 0) {
        // =========================================================================== mfma role
        // The two waves of a SIMD take ALTERNATE products: wave parity pw computes the products i = pw, pw + 2, ... (product
        // i = block row b(i), consumed by the chain in interval i) over ALL blocks of its SIMD; while it stores, waits and
        // refreshes, its partner multiplies.
        const int sm = wave < 4 ? wave - 1 : wave - 5;     // SIMD of the pair (waves w and w + 4 share one)
        const int pw = wave < 4 ? 0 : 1;
        const int mw = wave < 4 ? wave - 1 : wave - 2;     // progress word
        const RqOwn own = rq_own(NB, CS, sm);
        v2d_ arP[2 * RQ_PFU];
        long long qc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tq = 0;
#define QTICK(slot) if (PROF && a.prof && (wave == 2 || wave == 6)) { long long now_ = (long long)__builtin_amdgcn_s_memtime(); qc[slot] += now_ - tq; tq = now_; }
        // No address arithmetic and no LDS traffic in the product loop.  Unit u of this SIMD is block sm + 3 u:
        //   B operands: PERSISTENT in registers (4 per unit); a product only re-reads the (at most two) blocks committed
        //   since this wave's previous product;
        //   A fragments of block row `row`: buffer loads, descriptor = Apack2, scalar offset = row * KS * 512 + block * 2048
        //   (one s_add per unit), vector offset = lane * 16, through a ring of RQ_PFU units.  Units past the owned ones
        //   read the next block row, or zeros past the end of the buffer: loaded, never used.
        const unsigned vlane = (unsigned)lane * 16u;
        const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(Apack2), 0, NB * KS * 512, 0x00020000);
        // (three opaque LDS bases 8 units apart keep every operand read within the 16-bit immediate of ds_read: no
        //  per-unit address registers)
        typedef __attribute__((address_space(3))) const double rq_lds_cd;
        rq_lds_cd *xbase = (rq_lds_cd *)(Xs + (lane >> 4) * 16 + (lane & 15) + sm * 256);
        rq_lds_cd *xb3[3] = {xbase, xbase + 8 * 12 * 64, xbase + 16 * 12 * 64};
        asm volatile("" : "+v"(xb3[0]), "+v"(xb3[1]), "+v"(xb3[2]));
        const int rowstride = KS * 512;
        typedef unsigned rq_u4 __attribute__((ext_vector_type(4)));
#define RQ_LDA(dst, soff, vo) { const rq_u4 t0_ = __builtin_amdgcn_raw_buffer_load_b128(arsrc, (vo), (soff), 0);           \
                                const rq_u4 t1_ = __builtin_amdgcn_raw_buffer_load_b128(arsrc, (vo) + 1024u, (soff), 0);   \
                                (dst)[0] = __builtin_bit_cast(v2d_, t0_); (dst)[1] = __builtin_bit_cast(v2d_, t1_); }
        double bq[4 * RQ_PERS], bx[RQ_MAXU > RQ_PERS ? 4 * (RQ_MAXU - RQ_PERS) : 4];
#pragma unroll
        for (int u = 0; u < RQ_PERS; u++)
#pragma unroll
            for (int q = 0; q < 4; q++) bq[4 * u + q] = xb3[u >> 3][(12 * (u & 7) + q) * 64];
        int row = pw % NB;
#pragma unroll
        for (int U = 0; U < RQ_PFU; U++) RQ_LDA(arP + 2 * U, row * rowstride + (sm + RQ_NSIMD * U) * 2048, vlane)
        if (PROF && a.prof) tq = (long long)__builtin_amdgcn_s_memtime();
        for (int64_t i = pw; i < gmax; i += 2) {
            // block row of product i + 2 and the two blocks the chain supplies itself (being rewritten / rewritten last)
            int row2 = row + 2; row2 = row2 >= NB ? row2 - NB : row2;
            const int h1 = (i >= 1) ? (row == 0 ? NB - 1 : row - 1) : -1;
            const int h2 = (i >= 2) ? (h1 == 0 ? NB - 1 : h1 - 1) : -1;
            // ... and the two committed since this wave's previous product (its holes then): their operands are stale
            const int r1 = (i >= 3) ? (h2 == 0 ? NB - 1 : h2 - 1) : -1;
            const int r2 = (i >= 4) ? (r1 == 0 ? NB - 1 : r1 - 1) : -1;
            // units this product leaves out: the two holes (when this SIMD owns them) and everything past the owned units
            unsigned skip = ~0u << own.nu, fresh = 0u;
            if (h1 >= 0 && h1 % RQ_NSIMD == sm && h1 >= own.first) skip |= 1u << ((h1 - own.first) / RQ_NSIMD);
            if (h2 >= 0 && h2 % RQ_NSIMD == sm && h2 >= own.first) skip |= 1u << ((h2 - own.first) / RQ_NSIMD);
            if (r1 >= 0 && r1 % RQ_NSIMD == sm && r1 >= own.first) fresh |= 1u << ((r1 - own.first) / RQ_NSIMD);
            if (r2 >= 0 && r2 % RQ_NSIMD == sm && r2 >= own.first) fresh |= 1u << ((r2 - own.first) / RQ_NSIMD);
            fresh &= ~(~0u << own.nu);
            const int so1 = row * rowstride + sm * 2048, so2 = row2 * rowstride + sm * 2048;
            QTICK(0)
            RQ_TRACE(i, 0)
            bool stop = false;
            if (i >= 3) {
                // every block except the two holes must be final: the latest one was committed in interval i - 3
                for (;;) {
                    const rq_i4 s4 = rq_sync_read(sy);
                    if (s4[RQ_STOP]) { stop = true; break; }
                    if (s4[RQ_COMMIT] >= (int)i - 2) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (stop) break;
                RQ_TRACE(i, 1)
                if (fresh) {
#pragma unroll
                    for (int u = 0; u < RQ_PERS; u++)
                        if ((fresh >> u) & 1u) {   // wave-uniform
#pragma unroll
                            for (int q = 0; q < 4; q++) bq[4 * u + q] = xb3[u >> 3][(12 * (u & 7) + q) * 64];
                        }
                }
            }
#pragma unroll
            for (int u = RQ_PERS; u < RQ_MAXU; u++)
#pragma unroll
                for (int q = 0; q < 4; q++) bx[4 * (u - RQ_PERS) + q] = xb3[u >> 3][(12 * (u & 7) + q) * 64];
            QTICK(1)
            RQ_TRACE(i, 2)
            v4d_ acc = {0.0, 0.0, 0.0, 0.0}, acc1 = acc;    // two chains: a wave issues an MFMA every >= 64 cycles anyway
#pragma unroll
            for (int third = 0; third < RQ_RND; third++) {
#pragma unroll
                for (int U = 0; U < RQ_PFU; U++) {
                    const int u = RQ_PFU * third + U;
                    if (u < RQ_MAXU && !((skip >> u) & 1u)) {   // wave-uniform
                        const double *bu = u < RQ_PERS ? bq + 4 * u : bx + 4 * (u < RQ_MAXU ? u - RQ_PERS : 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arP[2 * U][0], bu[0], acc, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(arP[2 * U][1], bu[1], acc1, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arP[2 * U + 1][0], bu[2], acc, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(arP[2 * U + 1][1], bu[3], acc1, 0, 0, 0);
                    }
                    // unconditional refill of the ring slot: the unit RQ_PFU further on, then the first units of row2
                    if (third < RQ_RND - 1) { if (u + RQ_PFU < RQ_MAXU) RQ_LDA(arP + 2 * U, so1 + RQ_NSIMD * 2048 * (u + RQ_PFU), vlane) }
                    else RQ_LDA(arP + 2 * U, so2 + RQ_NSIMD * 2048 * U, vlane)
                }
            }
            acc = acc + acc1;
            QTICK(2)
            RQ_TRACE(i, 3)
            if (i >= 2) {
                // the slot of this wave (parity) held product i - 2: read by the chain at the start of interval i - 2
                for (;;) {
                    const rq_i4 s4 = rq_sync_read(sy);
                    if (s4[RQ_STOP]) { stop = true; break; }
                    if (s4[RQ_CONS] >= (int)i - 1) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (stop) break;
            }
            QTICK(3)
            RQ_TRACE(i, 4)
            {
                double *part = part2 + (int)(i & 1) * RQ_NSIMD * 256 + sm * 256;
#pragma unroll
                for (int v = 0; v < 4; v++) part[v * 64 + (lane & 15) * 4 + (lane >> 4)] = acc[v];
            }
            RQ_TRACE(i, 5)
            rq_sync_write(sy, RQ_PARTS + mw, (int)i + 1, lane);
            RQ_TRACE(i, 6)
            row = row2;
        }
        if (PROF && a.prof && (tid == 128 || tid == 384))   // the elder and the younger wave of one pair
            for (int k = 0; k < 4; k++) a.prof[tile * 16 + (tid == 128 ? 8 : 12) + k] = qc[k];
#undef QTICK
#undef RQ_LDA
    } else {
        // ========================================================================== chain role
        __builtin_amdgcn_s_setprio(3);
        const int r = lane >> 2, gq = lane & 3;
        const int64_t gr = tile * 16 + r;
        const bool live_r = gr < a.R;
        // feasible set of this lane's restart (registers for the whole kernel): [-symb, -syma] u [syma, symb]
        const int Un = TC.n[r], Uslow = TC.slow[r];
        const double Ul0 = TC.lo[r], Uh0 = TC.hi[r], Ul1 = TC.lo[16 + r], Uh1 = TC.hi[16 + r];
        const bool two = Un >= 2;
        const double thr = two ? 1e-7 * (Ul1 - Uh0) : 0.0;
        const double syma = two ? Ul1 : 0.0, symb = two ? Uh1 : Uh0;
        ChainState S;
        S.fcur = 0.0; S.upd_counter = 0; S.visits = 0; S.accepted = 0; S.sweeps = 0;
        S.conv = true; S.status = 0;
        if (live_r) S.conv = a.flag[gr] ? false : true;
        // tracked objective: the four lanes of a quad hold partial sums; lane g == 0 starts from the evaluated value
        double fpart = (live_r && gq == 0) ? a.f0cur[gr] : 0.0;
        const RqOwn cown = rq_own(NB, CS, RQ_NSIMD);
        v2d_ arC[2 * CSU];
        double bqC[4 * CSU];
        double afix[4] = {0.0, 0.0, 0.0, 0.0}, afix2[4] = {0.0, 0.0, 0.0, 0.0};
        v4d_ carry = {0.0, 0.0, 0.0, 0.0};      // the block rewritten last times the fragments of the row after next
        if (CS > 0) {
            rq_load_A<CSU>(arC, Apack2, KS, cown, lane, 0);
            rq_load_B<CSU>(bqC, Xs, cown, lane);
            // prologue share: product of block row 0 over the chain's blocks (no hole), into the chain's plane
            v4d_ acc = rq_product<CSU>(arC, bqC, Apack2, KS, cown, lane, -1, -1, 1, v4d_{0.0, 0.0, 0.0, 0.0});
#pragma unroll
            for (int v = 0; v < 4; v++) fixp[v * 64 + (lane & 15) * 4 + (lane >> 4)] = acc[v];
        } else {
#pragma unroll
            for (int v = 0; v < 4; v++) fixp[v * 64 + lane] = 0.0;
        }
        long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = 0;
#define PROF_TICK(slot) if (PROF && a.prof) { long long now_ = (long long)__builtin_amdgcn_s_memtime(); pc[slot] += now_ - tp; tp = now_; }
        if (PROF && a.prof) tp = (long long)__builtin_amdgcn_s_memtime();
        const double tolv = a.tol;
        int b = 0;
        int64_t t = 0;
        for (int64_t g = 0; g < gmax; g++) {
            const int bn = (b + 1 == NB) ? 0 : b + 1, bn2 = (bn + 1 == NB) ? 0 : bn + 1;
            const int bprev = (b == 0) ? NB - 1 : b - 1;
            const int cur = (int)(g & 1);
            const double *DU = DU2 + cur * 256, *rtb = rtb2 + cur * 16, *dgb = dg2 + cur * 16, *hqb = hqb2 + cur * 16;
            const double *part = part2 + cur * RQ_NSIMD * 256;
            PROF_TICK(0)
            RQ_TRACE(g, 0)
            // partial tiles (product g: the three waves whose turn it was) and staged operands of block b
            for (;;) {
                const rq_i4 p4 = rq_sync_read(sy + RQ_PARTS), p2 = rq_sync_read(sy + RQ_PARTS + 4);
                int lo4 = p4[0] < p4[1] ? p4[0] : p4[1];
                const int lo2 = p4[2] < p4[3] ? p4[2] : p4[3], lo1 = p2[0] < p2[1] ? p2[0] : p2[1];
                lo4 = lo4 < lo2 ? lo4 : lo2;
                lo4 = lo4 < lo1 ? lo4 : lo1;
                // an mfma wave publishes i + 1 after product i and only computes every other product: "all six >= g" says the
                // three waves of parity g have delivered product g (their values jump by 2)
                if (lo4 >= (int)g && p2[2] >= (int)g + 1) break;
                __builtin_amdgcn_s_sleep(1);
            }
            PROF_TICK(1)
            RQ_TRACE(g, 1)
            // ---- G + q/2 of the lane's own columns: its own plane, then the three partial tiles, in a fixed order, then q/2
            double gb[4], g0[4], xo[4], xn[4], rto[4], t2o[4];
#pragma unroll
            for (int v = 0; v < 4; v++) {
                double s = fixp[v * 64 + lane];
#pragma unroll
                for (int w = 0; w < RQ_NSIMD; w++) s += part[w * 256 + v * 64 + lane];
                s += hqb[4 * v + gq];
                gb[v] = s;
                g0[v] = s;
            }
            rq_sync_write(sy, RQ_CONS, (int)g + 1, lane);     // the partial tiles have been read (LDS is in order per wave)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                xo[v] = Xs[(16 * b + 4 * v + gq) * 16 + r];
                rto[v] = rtb[4 * v + gq];
                t2o[v] = dgb[4 * v + gq];
            }
            {   // A fragments of this block's k-steps in the next two block rows (the chain's contribution to both)
                const double *ap = Apack + ((int64_t)bn * KS + 4 * b) * 64 + lane;
                const double *ap2 = Apack + ((int64_t)bn2 * KS + 4 * b) * 64 + lane;
#pragma unroll
                for (int u = 0; u < 4; u++) { afix[u] = ap[u * 64]; afix2[u] = ap2[u * 64]; }
            }
            if (b == 0 && !S.conv) S.sweeps++;
            PROF_TICK(2)
            const bool act = !S.conv;
            const bool actn = act && Un > 0;
            const double tole = actn ? tolv : QM_INF;     // a restart that is not sweeping never moves
            // ---- the 16 steps: only what the next step waits for
#pragma unroll
            for (int c = 0; c < 16; c++) {
                const int v = c >> 2, go = c & 3;
                // every lane works on its own column 4 v + gq; only the owner quad-lane (gq == go) is at step c
                const double xv = __builtin_fma(-gb[v], rto[v], xo[v]);              // vertex of the scalar objective
                const double pick = __builtin_copysign(fmin(fmax(fabs(xv), syma), symb), xv);
                const double dlt = pick - xo[v];
                const double dl = (fabs(dlt) > tole) ? dlt : 0.0;
                double delta;
                if (go == 0) delta = rq_quad_bcast<0x00>(dl);
                else if (go == 1) delta = rq_quad_bcast<0x55>(dl);
                else if (go == 2) delta = rq_quad_bcast<0xAA>(dl);
                else delta = rq_quad_bcast<0xFF>(dl);
                // fold the move into the columns the lane owns that are still ahead (the masked block has zeros elsewhere,
                // in particular at the lane's own finished columns: their G stays what the decision saw)
#pragma unroll
                for (int v2 = v; v2 < 4; v2++) gb[v2] = __builtin_fma(DU[c * 16 + 4 * v2 + gq], delta, gb[v2]);
            }
            PROF_TICK(3)
            // ---- once per block, per own column: the decision again from the frozen G (bit-identical to what the step
            // computed when the lane was the owner), new x, near-tie test, move mask, objective tracking
            bool allfar = true;
            unsigned mv = 0;
            double fadd = 0.0;
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const double xv = __builtin_fma(-gb[v], rto[v], xo[v]);
                const double pick = __builtin_copysign(fmin(fmax(fabs(xv), syma), symb), xv);
                const double dlt = pick - xo[v];
                const bool mvd = fabs(dlt) > tole;
                const double d = mvd ? dlt : 0.0;
                xn[v] = mvd ? pick : xo[v];
                allfar = allfar && (fabs(xv) > thr);                                   // false for NaN as well
                mv |= mvd ? (1u << (4 * v + gq)) : 0u;
                // f(x + d e_i) - f(x) = d (2 (P x)_i + q_i + P_ii d) = d (t2 d + 2 g):  g = G_i + q_i / 2 contains P_ii x_i
                fadd = __builtin_fma(d, __builtin_fma(t2o[v], d, gb[v] + gb[v]), fadd);
            }
            mv = rq_quad_or(mv);                                                       // bit c = coordinate c moved
            bool redo = act && Un > 0 && (!allfar || Uslow != 0);
            if (__builtin_amdgcn_ballot_w64(redo) == 0ull) {
                if (act) {
                    fpart += fadd;
                    const int accn = __builtin_popcount(mv);
                    const int upd = mv ? (__builtin_clz(mv) - 16) : (int)S.upd_counter + 16;
                    S.accepted += accn;
                    const int over = upd - (int)P.n;
                    S.visits += 16 - (over > 0 ? over : 0);
                    S.upd_counter = upd;
                    if (over >= 0) S.conv = true;
                }
#pragma unroll
                for (int v = 0; v < 4; v++) Xs[(16 * b + 4 * v + gq) * 16 + r] = xn[v];
            } else {
                // ---- generic loop (rare): the reference's arithmetic; G tile (kept from the block's start: the partial
                // tiles may already be overwritten) rebuilt in the chain's plane, all four lanes of a quad walk their
                // restart redundantly (same values, benign identical LDS writes)
                if (PROF) pc[6]++;
                S.fcur = rq_quad_sum(fpart);
                double *Gsc = fixp;
#pragma unroll
                for (int v = 0; v < 4; v++) Gsc[(4 * v + gq) * 16 + r] = g0[v];
                for (int c = 0; c < 16; c++) {
                    const int64_t i = 16 * (int64_t)b + c;
                    FeasSet<MAXC> C;
                    C.n = Un; C.lo[0] = Ul0; C.hi[0] = Uh0; C.lo[1] = Ul1; C.hi[1] = Uh1;
                    const double t2g = dgb[c];
                    const double xi = Xs[i * 16 + r];
                    const double hq = hqb[c];
                    const double t1 = 2.0 * ((Gsc[c * 16 + r] - hq) - t2g * xi) + (hq + hq);
                    const double t0 = S.fcur - xi * (t2g * xi + t1);
                    DrawKey dk{a.seed, a.first_index + (uint64_t)gr, (uint32_t)i, (uint32_t)t | 0x80000000u, 0u};
                    double xnew = xi;
                    int got = S.conv ? 0 : onevar_minimise<MAXC>(t2g, t1, t0, C, dk, &xnew);
                    bool moved;
                    double delta;
                    chain_commit<MAXC>(S, got, xnew, xi, t2g, t1, t0, a.tol, P.n, moved, delta);
                    if (moved) {
                        Xs[i * 16 + r] = xnew;
                        for (int c2 = c + 1; c2 < 16; c2++) Gsc[c2 * 16 + r] += DU[c * 16 + c2] * delta;
                    }
                }
                fpart = (gq == 0) ? S.fcur : 0.0;
            }
            rq_sync_write(sy, RQ_COMMIT, (int)g + 1, lane);   // block b is in the X tile; its staged operands are free
            PROF_TICK(4)
            RQ_TRACE(g, 2)
            const unsigned long long livem = __builtin_amdgcn_ballot_w64(!S.conv);
            if (PROF) pc[5]++;
            if (livem == 0ull || g + 1 >= gmax) break;
            // ---- the chain's part of the next products: the block just committed times the fragments of the next TWO block
            // rows (the mfma waves leave out the last two blocks rewritten: none of them ever waits for a fresh commit), and
            // its own share of the next row
            {
                v4d_ acc = carry, acc2 = {0.0, 0.0, 0.0, 0.0};
                const int xoff = (4 * b) * 64 + (lane >> 4) * 16 + (lane & 15);
                double xb4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) xb4[u] = Xs[xoff + u * 64];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afix[u], xb4[u], acc, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(afix2[u], xb4[u], acc2, 0, 0, 0);
                }
                carry = acc2;
                if (CS > 0) {
                    const int us = rq_slot(cown, b);
                    if (us >= 0) rq_refresh_B<CSU>(bqC, Xs, cown, lane, us);
                    acc = rq_product<CSU>(arC, bqC, Apack2, KS, cown, lane, us, (g > 0 ? rq_slot(cown, bprev) : -1), bn2, acc);
                }
#pragma unroll
                for (int v = 0; v < 4; v++) fixp[v * 64 + (lane & 15) * 4 + (lane >> 4)] = acc[v];
            }
            PROF_TICK(7)
            RQ_TRACE(g, 3)
            b = bn;
            if (b == 0) t++;
        }
        rq_sync_write(sy, RQ_STOP, 1, lane);
        const double ftot = rq_quad_sum(fpart);
        if (gq == 0 && live_r) {
            a.visits[gr] = S.visits; a.accepted[gr] = S.accepted; a.sweeps[gr] = S.sweeps;
            a.status[gr] = S.status;
            if (a.f0out && a.flag[gr]) a.f0out[gr] = ftot;   // tracked exactly through every accepted move
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
    for (int64_t idx = tid; idx < n16 * 16; idx += 512) Xg[idx] = Xs[idx];
}

}  // namespace qcqpmi
