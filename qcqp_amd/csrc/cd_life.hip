// cd_life_kernel -- improve_coord_descent (qcqp.py:181-192) for a queue of restarts inside one persistent launch, second
// generation (round 5).  What it computes, restart by restart, is what cd_phase2_qs_kernel<CS, LIFE = true> (cd_queue.hip)
// computes -- suggest(RANDOM) / resident start, phase 1 through cd_phase1_sep.h, gate, the blocked Gauss-Seidel phase 2 of
// cd_phase2_q.h with its near-tie replay in the reference's arithmetic, the objective from the converged window -- and the
// EPISODE structure is the same (16 slots per workgroup; at a sweep boundary finished slots are written out and refilled from
// the queue; products are always summed in one association so that results do not depend on the episode boundaries).
// What is new is the layout of the work on the chip (cd_life.h): four-wave workgroups, two per CU, roles by hardware SIMD,
// the X tile in a private global tile with a ring of the four blocks committed last in LDS, every multiplying wave
// computes EVERY product over its own blocks of the contraction, the chain wave stages its own small operands; plus the
// generalisations: n not a multiple of 16, 1024 < n <= 2048 with six multiplying waves, step kinds for single classes
// with up to two intervals and for a zero diagonal.
#include "cd_life.h"

#include <cstdio>
#include <cstdlib>

#include "onevar.h"
#include "cd_phase1_sep.h"

namespace qcqpmi {
typedef double v4d __attribute__((ext_vector_type(4)));
template <typename XPtr>
__device__ inline v4d block_rows_times_X(const double *__restrict__ Ab, XPtr Xs, int kk0, int kk1, int lane, v4d acc) {
    const int xoff = (lane >> 4) * 16 + (lane & 15);
    for (int kk = kk0; kk < kk1; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ab[(int64_t)kk * 64 + lane], Xs[kk * 64 + xoff], acc, 0, 0, 0);
    return acc;
}
}  // namespace qcqpmi

#include "cd_phase2_q.h"      // rq_quad_*, RqOwn, rq_block / rq_slot, the LDS flag protocol, RQ_PFU / RQ_RND / RQ_PERS / RQ_MAXU

namespace qcqpmi {
namespace {

#define LG __attribute__((address_space(1)))
template <class T>
__device__ __attribute__((always_inline)) inline LG T *l2_g(T *p) { return (LG T *)p; }
__device__ inline int l2_load_int(LG const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline int l2_add(LG int *p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ inline unsigned long long l2_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double l2_unkey(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
__device__ inline double l2_wave_max(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const double w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}

// the scalar code of a restart's start -- same expressions as cd_queue.hip (which has to CALL them: inlined they spilled the
// product loop of its multiplying waves; here the roles are functions of their own and the kernel body is free to inline)
__device__ __attribute__((always_inline)) inline double l2_keyed_normal_pair(uint64_t seed, uint64_t restart, uint64_t elem, double *odd) {
    const U4 o = philox4x32_10((uint32_t)(elem >> 1), (uint32_t)(elem >> 33), 0xA5A50000u, (uint32_t)restart, (uint32_t)seed,
                               (uint32_t)(seed >> 32) ^ (uint32_t)(restart >> 32));
    const double u1 = (((double)(o.x >> 5) * 67108864.0 + (double)(o.y >> 6)) + 0.5) / 9007199254740992.0;
    const double u2 = u53(o.z, o.w);
    const double rad = sqrt(-2.0 * log(u1));
    const double ang = 6.283185307179586476925286766559 * u2;
    *odd = rad * sin(ang);
    return rad * cos(ang);
}
__device__ __attribute__((always_inline)) inline double l2_p1_visit(double p, double q, double r, int relop, int64_t i, double x, double tol,
                                                        double viol_tol, uint64_t seed, uint64_t restart, int64_t t, int *flags,
                                                        double *vafter) {
    P1Visit V;
    if (q == 0.0 && relop == RELOP_EQ && p > 1e-4) {
        p1_band_visit(p, q, r, i, x, tol, viol_tol, seed, restart, t, V);
    } else {
        const double cp[1] = {p}, cq[1] = {q}, cr[1] = {r};
        const int crel[1] = {relop};
        p1_sep_visit_core<1>(1, cp, cq, cr, crel, i, x, tol, viol_tol, seed, restart, t, V);
    }
    *flags = (V.moved ? 1 : 0) | ((-V.status) << 8);
    *vafter = V.vafter;
    return x;
}
constexpr int L2_WD = 1 << 21;        // polls before a wait gives up (a wait is a few hundred polls at most)
enum { L2_ABORT = 3 };                // sync word: a watchdog fired in this workgroup

// blocks of the contraction owned by multiplying wave m of NMW (the chain wave keeps the last CS blocks)
__device__ __attribute__((always_inline)) inline RqOwn l2_own(int NB, int CS, int m, int NMW) {
    RqOwn o;
    const int rest = NB - CS;
    o.NB = NB;
    if (m < 0) { o.first = rest; o.stride = 1; o.nu = CS; }
    else { o.first = m; o.stride = NMW; o.nu = m < rest ? (rest - m + NMW - 1) / NMW : 0; }
    return o;
}

// chain share: A fragments (pair-packed) of the owned blocks for block row bn -> registers
template <int NU>
__device__ __attribute__((always_inline)) inline void l2_load_A(v2d_ (&ar)[2 * NU], LG const double *Apack2, int KS, const RqOwn &o, int lane, int bn) {
#pragma unroll
    for (int U = 0; U < NU; U++) {
        LG const v2d_ *ap = (LG const v2d_ *)Apack2 + ((int64_t)bn * (KS / 2) + 2 * rq_block(o, U)) * 64;
        ar[2 * U] = ap[(unsigned)lane];
        ar[2 * U + 1] = ap[64u + (unsigned)lane];
    }
}
template <int NU>
__device__ __attribute__((always_inline)) inline v4d_ l2_product(v2d_ (&ar)[2 * NU], const double (&bq)[4 * NU], LG const double *Apack2, int KS,
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
        {
            LG const v2d_ *ap = (LG const v2d_ *)Apack2 + ((int64_t)bn2 * (KS / 2) + 2 * rq_block(o, U)) * 64;
            ar[2 * U] = ap[(unsigned)lane];
            ar[2 * U + 1] = ap[64u + (unsigned)lane];
        }
    }
    return (acc + acc1) + (acc2 + acc3);
}

// ---- dynamic LDS of a workgroup (doubles), the same view in the kernel and in the role functions.  Shared words first:
//   ctl 8, simdof 8 (ints: 16), p1cols 16 (ints: 32), jn 8 (ints: the two chains' joint decisions, TILES = 2)
// then one block PER TILE (a workgroup runs TILES = 1 or 2 tiles of 16 slots; a role works on the block of its tile):
//   part2   2 NMW 256   partial G tiles (one per multiplying wave), [v][4 r + g], by product parity
//   fixp    256         the chain wave's own plane (fix-up + its share)
//   gtile   256         G + q/2 of the block as the chain found it, [c][r]: the generic path's tile
//   DU2     2 x 256     strictly upper triangle of the diagonal block, by interval parity
//   sc2     2 x 48      per parity: q0 / 2 [16], 1 / P0[i,i] [16], P0[i,i] [16]
//   ring    4 x 256     the four blocks committed last, [j][r] (= MFMA B layout k-step by k-step)
//   cshare  CS x 256    the blocks of the chain wave's own share of the contraction, same layout
//   slot tables (slack, feasible set, restart id, new / finished flags, outputs), sy: the synchronisation words,
//   cst: the chain wave's per-lane state between episodes [field][lane], phase-1 words, par
//   LR (factored objective, see l2_mfma_lr_role): ytile L2_YBMAX x 256 -- Y = L^T X of the tile's 16 slots between episodes, [row][slot];
//   pend 2 x 256 -- the moves of the two blocks a sweep rewrites last, [j][r], until the next episode has applied them
//   TC: the slots' feasible sets [interval][class][slot] (one class; L2_KCL classes of up to three intervals for the multi-class
//   kinds), clsb 2 x 16 ints: the classes of the staged block's coordinates
constexpr int L2_SHARED_DOUBLES = 8 + 8 + 16 + 8;
constexpr int L2_YBMAX = 18;          // blocks of 16 rows of Y a tile can hold: factor rank <= 288
constexpr int L2_YU = 6;              // ... per multiplying wave (three per tile)
constexpr int L2_KCL = 4;            // constraint classes of the multi-class kinds (GENK / LINK); up to two constraints per coordinate
constexpr int l2_tile_doubles(int nmw, int csu, int lr, int kcl = 1, int maxc = 1) {
    return 2 * nmw * 256 + 256 + 256 + 2 * 256 + 2 * 48 + 4 * 256 + csu * 256 + 16 + 8 + 16 * 4 + 8 * 5 + 64 * 10 + 16 * 3 + 8 * 5 + 32 +
           (lr ? L2_YBMAX * 256 + 2 * 256 : 0) +
           2 * (maxc + 1) * 16 * kcl + 2 * 8 * kcl + (kcl > 1 ? 16 : 0);      // the feasible-set table (per class), the staged class ids
}
#define L2_LDS_VIEW(tile_) \
    extern __shared__ double smem[]; \
    int *ctl = (int *)smem; \
    int *simdof = (int *)(smem + 8); \
    int *p1cols = (int *)(smem + 16); \
    int *jn = (int *)(smem + 32); \
    double *sp = smem + L2_SHARED_DOUBLES + (tile_) * l2_tile_doubles(NMW, CSU, LRV, KCLV, MAXC); \
    double *part2 = sp; sp += 2 * NMW * 256; \
    double *fixp = sp; sp += 256; \
    double *gtile = sp; sp += 256; \
    double *DU2 = sp; sp += 2 * 256; \
    double *sc2 = sp; sp += 2 * 48; \
    double *ring = sp; sp += 4 * 256; \
    double *cshare = sp; sp += CSU * 256; \
    double *slk = sp; sp += 16; \
    rq_lds_int *sy = (rq_lds_int *)(int *)sp; sp += 8; \
    double *of0 = sp; sp += 16; \
    long long *ovis = (long long *)sp; sp += 16; \
    long long *oacc = (long long *)sp; sp += 16; \
    long long *oswp = (long long *)sp; sp += 16; \
    int *sid = (int *)sp; sp += 8; \
    int *snew = (int *)sp; sp += 8; \
    int *sfin = (int *)sp; sp += 8; \
    int *ost = (int *)sp; sp += 8; \
    int *gatep = (int *)sp; sp += 8; \
    long long *cst = (long long *)sp; sp += 64 * 10; \
    unsigned long long *sseed = (unsigned long long *)sp; sp += 16; \
    unsigned long long *sfirst = (unsigned long long *)sp; sp += 16; \
    unsigned long long *p1key = (unsigned long long *)sp; sp += 16; \
    int *p1upd = (int *)sp; sp += 8; \
    int *p1fin = (int *)sp; sp += 8; \
    int *p1sw = (int *)sp; sp += 8; \
    int *p1st = (int *)sp; sp += 8; \
    L2Par *par = (L2Par *)sp; sp += 32;      /* (sizeof(L2Par) <= 256 bytes: static_assert below) */ \
    double *ytile = sp; sp += LRV ? L2_YBMAX * 256 : 0; \
    double *pend = sp; sp += LRV ? 2 * 256 : 0; \
    SetTable<MAXC> TC;       /* last: the roles that never look at it (MAXC = KCLV = 1 there) see every other offset unchanged */ \
    TC.slots = 16 * KCLV; \
    TC.lo = sp; sp += (MAXC + 1) * 16 * KCLV; \
    TC.hi = sp; sp += (MAXC + 1) * 16 * KCLV; \
    TC.n = (int *)sp; sp += 8 * KCLV; \
    TC.slow = (int *)sp; sp += 8 * KCLV; \
    int *clsb = (int *)sp;   /* [2][16] class of the coordinates of the staged block, by interval parity (multi-class kinds) */
// the same per-slot array of the other tile (every tile block has the same layout)
template <class T>
__device__ __attribute__((always_inline)) inline T *l2_tl(T *p, int t, int tile_doubles) { return (T *)((double *)p + t * tile_doubles); }

// parameters the role functions need, in LDS: arguments of a real function call travel in vector registers, i.e. the callee
// would have to treat NB, the base pointers, ... as lane-dependent (waterfall loops around every buffer descriptor);
// read from LDS and made wave-uniform explicitly they are scalars again
struct L2Par {
    const double *Apack, *Apack2, *Dpack, *Spack;
    const double *Gpack, *Upack;          // LR: fragments of the factor for the products / the updates of Y (cd_life.h)
    int RB, pad_;                         // LR: blocks of 16 rows of Y
    double *Xg;
    int *next;
    unsigned long long *prof;
    long long num_iters, n16;
    double tol, r0, fbound;
    int NB, KS, n, nlast, Rtotal, dbg;
    const int *cls;                       // [n16] class of each coordinate (multi-class kinds)
};
static_assert(sizeof(L2Par) <= 32 * sizeof(double), "L2Par outgrew its LDS slot");
__device__ __attribute__((always_inline)) inline int l2_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __attribute__((always_inline)) inline long long l2_uni(long long v) {
    const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((long long)hi << 32) | (long long)(unsigned)lo;
}
__device__ __attribute__((always_inline)) inline double l2_uni(double v) { return __longlong_as_double(l2_uni(__double_as_longlong(v))); }
template <class T>
__device__ __attribute__((always_inline)) inline T *l2_uni(T *p) { return (T *)(uintptr_t)l2_uni((long long)(uintptr_t)p); }

// =========================================================================== multiplying role (a real function: its
// register allocation -- 160 registers of persistent B operands, the A ring, two accumulators -- is its own; inlined into the
// kernel beside the chain role the two fought over the 256 registers and over the scalar file, and the product loop spilled
// whenever anything else in the kernel changed)
template <int NMW, int CS>
__device__ __attribute__((noinline)) void l2_mfma_role(int m_in, int t_in) {
    constexpr int MAXC = 1;
    constexpr int KCLV = 1;
    constexpr int CSU = CS > 0 ? CS : 1;
    constexpr int LRV = 0;
    const int tile = l2_uni(t_in);                 // the tile of the workgroup this wave multiplies for
    L2_LDS_VIEW(tile)
    (void)ytile; (void)pend; (void)clsb;
    (void)fixp; (void)gtile; (void)DU2; (void)sc2; (void)cshare; (void)slk; (void)TC; (void)of0; (void)ovis; (void)oacc; (void)oswp; (void)sid; (void)snew;
    (void)sfin; (void)ost; (void)ctl; (void)cst; (void)sseed; (void)sfirst; (void)p1key; (void)p1upd; (void)p1fin; (void)p1sw; (void)p1st; (void)gatep; (void)simdof; (void)p1cols; (void)jn;
    const int lane = threadIdx.x & 63;
    const int m = l2_uni(m_in);
    const double *pApack2 = l2_uni(par->Apack2);
    double *pXg = l2_uni(par->Xg);
    unsigned long long *pprof = l2_uni(par->prof);
    const int NB = l2_uni(par->NB), KS = l2_uni(par->KS);
    const int64_t n16 = l2_uni(par->n16);
    const int rowmask = (l2_uni(par->dbg) & 1) ? 7 : -1;       // timing experiment: alias the block rows (results invalid)
    const int64_t gmax = (int64_t)1 << 40;         // the role ends through RQ_STOP
    // =========================================================================== multiplying role
    // Wave m owns the blocks m, m + NMW, ... of the contraction (unit u = block m + NMW u) and computes EVERY product
    // (product i = block row i mod NB, consumed by the chain in interval i) over them:
    //   B operands: PERSISTENT in registers (4 per unit), loaded from the global tile at the start of the episode; a
    //   product re-reads only the block committed three intervals ago (out of the LDS ring) -- the other two blocks
    //   rewritten since are the "holes" the chain supplies itself;
    //   A fragments of block row `row`: buffer loads, descriptor = P.Apack2, scalar offset = row KS 512 + block 2048,
    //   vector offset = lane 16, through a ring of RQ_PFU units refilled in place.
    const RqOwn own = l2_own(NB, CS, m, NMW);
    const int nu = own.nu;                         // units this wave owns
    v2d_ arP[2 * RQ_PFU];
    const unsigned vlane = (unsigned)lane * 16u;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(pApack2), 0, NB * KS * 512, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(pXg, 0, (int)(n16 * 128), 0x00020000);
    const int rowstride = KS * 512;
    typedef unsigned rq_u4 __attribute__((ext_vector_type(4)));
    typedef unsigned rq_u2 __attribute__((ext_vector_type(2)));
    #define L2_LDA(dst, soff, vo) { const rq_u4 t0_ = __builtin_amdgcn_raw_buffer_load_b128(arsrc, (vo), (soff), 0);           \
                            const rq_u4 t1_ = __builtin_amdgcn_raw_buffer_load_b128(arsrc, (vo) + 1024u, (soff), 0);   \
                            (dst)[0] = __builtin_bit_cast(v2d_, t0_); (dst)[1] = __builtin_bit_cast(v2d_, t1_); }
    double bq[4 * RQ_PERS];
    {
        const unsigned vl8 = (unsigned)lane * 8u;
    #pragma unroll
        for (int u = 0; u < RQ_PERS; u++)
    #pragma unroll
            for (int q = 0; q < 4; q++) {
                const rq_u2 t_ = __builtin_amdgcn_raw_buffer_load_b64(xrsrc, vl8, (m + NMW * u) * 2048 + q * 512, 0);
                bq[4 * u + q] = __builtin_bit_cast(double, t_);
            }
    }
    typedef __attribute__((address_space(3))) const double rq_lds_cd;
    rq_lds_cd *rbase = (rq_lds_cd *)(ring + lane);
    asm volatile("" : "+v"(rbase));
    int row = 0;
    #pragma unroll
    for (int U = 0; U < RQ_PFU; U++) L2_LDA(arP + 2 * U, row * rowstride + (m + ((U < nu) ? NMW * U : 0)) * 2048, vlane)
    int spins = 0;
    const bool prof_on = pprof != nullptr && m == 0 && tile == 0;
    long long pw_commit = 0, pw_cons = 0;
    for (int64_t i = 0; i < gmax; i++) {
        const int row2 = (row + 1 == NB) ? 0 : row + 1;
        // ALWAYS two holes, also for the first products of an episode (the chain supplies them from the tile as if the
        // sweep before had just ended): every product is summed in the same association wherever episodes begin
        const int h1 = row == 0 ? NB - 1 : row - 1;
        const int h2 = h1 == 0 ? NB - 1 : h1 - 1;
        const int r1 = (i >= 3) ? (h2 == 0 ? NB - 1 : h2 - 1) : -1;      // committed in interval i - 3: its operands are stale
        unsigned skip = ~0u << own.nu, fresh = 0u;
        if (h1 % NMW == m && h1 < NB - CS) skip |= 1u << (h1 / NMW);
        if (h2 % NMW == m && h2 < NB - CS) skip |= 1u << (h2 / NMW);
        if (r1 >= 0 && r1 % NMW == m && r1 < NB - CS) fresh |= 1u << (r1 / NMW);
        const int so1 = (row & rowmask) * rowstride + m * 2048, so2 = (row2 & rowmask) * rowstride + m * 2048;
        bool stop = false;
        if (i >= 3) {
            // every block except the two holes must be final: the latest one was committed in interval i - 3
            const long long tw0 = prof_on ? (long long)__builtin_amdgcn_s_memtime() : 0;
            for (;;) {
                const rq_i4 s4 = rq_sync_read(sy);
                if (s4[RQ_STOP]) { stop = true; break; }
                if (s4[RQ_COMMIT] >= (int)i - 2) break;
                if (++spins > L2_WD) { stop = true; rq_sync_write(sy, L2_ABORT, 1, lane); rq_sync_write(sy, RQ_STOP, 1, lane); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (stop) break;
            spins = 0;
            if (prof_on) pw_commit += (long long)__builtin_amdgcn_s_memtime() - tw0;
            if (fresh) {
                rq_lds_cd *rp = rbase + (int)((i - 3) & 3) * 256;
    #pragma unroll
                for (int u = 0; u < RQ_PERS; u++)
                    if ((fresh >> u) & 1u) {   // wave-uniform
    #pragma unroll
                        for (int q = 0; q < 4; q++) bq[4 * u + q] = rp[q * 64];
                    }
            }
        }
        v4d_ acc = {0.0, 0.0, 0.0, 0.0}, acc1 = acc;
    #pragma unroll
        for (int third = 0; third < RQ_RND; third++) {
    #pragma unroll
            for (int U = 0; U < RQ_PFU; U++) {
                const int u = RQ_PFU * third + U;
                if (u < RQ_MAXU && !((skip >> u) & 1u)) {   // wave-uniform
                    const double *bu = bq + 4 * u;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arP[2 * U][0], bu[0], acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(arP[2 * U][1], bu[1], acc1, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arP[2 * U + 1][0], bu[2], acc, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(arP[2 * U + 1][1], bu[3], acc1, 0, 0, 0);
                }
                // unconditional refill of the ring slot: the unit RQ_PFU further on, then the first units of the next row
                // (a unit the wave does not own: the fragment of its first unit again -- a line the vector cache holds, not another one
                //  from L2; making the loads conditional instead costs the counted waits of the ring: 48 -> 63 ms at n = 1024)
                if (third < RQ_RND - 1) { if (u + RQ_PFU < RQ_MAXU) L2_LDA(arP + 2 * U, so1 + ((u + RQ_PFU < nu) ? NMW * 2048 * (u + RQ_PFU) : 0), vlane) }
                else L2_LDA(arP + 2 * U, so2 + ((U < nu) ? NMW * 2048 * U : 0), vlane)
            }
        }
        acc = acc + acc1;
        if (i >= 2) {
            // this parity's slot held product i - 2: read by the chain at the start of interval i - 2
            const long long tw0 = prof_on ? (long long)__builtin_amdgcn_s_memtime() : 0;
            for (;;) {
                const rq_i4 s4 = rq_sync_read(sy);
                if (s4[RQ_STOP]) { stop = true; break; }
                if (s4[RQ_CONS] >= (int)i - 1) break;
                if (++spins > L2_WD) { stop = true; rq_sync_write(sy, L2_ABORT, 1, lane); rq_sync_write(sy, RQ_STOP, 1, lane); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (stop) break;
            spins = 0;
            if (prof_on) pw_cons += (long long)__builtin_amdgcn_s_memtime() - tw0;
        }
        {
            double *part = part2 + (int)(i & 1) * NMW * 256 + m * 256;
    #pragma unroll
            for (int v = 0; v < 4; v++) part[v * 64 + (lane & 15) * 4 + (lane >> 4)] = acc[v];
        }
        rq_sync_write(sy, RQ_PARTS + m, (int)i + 1, lane);
        row = row2;
    }
    if (prof_on && lane == 0) {
        atomicAdd(pprof + 9, (unsigned long long)pw_commit);
        atomicAdd(pprof + 10, (unsigned long long)pw_cons);
    }
    #undef L2_LDA
}


// =========================================================================== multiplying role, FACTORED objective (round 6)
// P0 = L L^T with L n x r, r <= 16 L2_YBMAX (Boolean least squares: P0 = A^T A, r = rows of A = n / 4): the product of a block row
// is  G_b = L[I_b, :] (L^T X)  and the tile's  Y = L^T X  (r x 16, 32 KB at r = 256 against the X tile's 128 KB) is CARRIED:
// after the chain has committed block b with the moves D_b = X_b(new) - X_b(old),  Y += L[I_b, :]^T D_b.  Per block interval that
// is 4 r / 16 MFMAs for the product and as many for the update -- 128 at r = 256 -- instead of the 256 - 8 of P0[I_b, :] X.
//   * wave m of the tile's three owns the blocks beta = m, m + 3, ... of 16 rows of Y, IN ITS ACCUMULATORS: the C layout of a
//     16 x 16 fp64 tile (lane l, element v: row (l >> 4) + 4 v, column l & 15) is the B layout of its four k-steps, so the same
//     registers are updated by one MFMA and multiplied by the next;
//   * association.  Product i (block row i mod NB) is taken with Y as of the commit of interval i - 3 -- the moves of the two
//     blocks rewritten since are the chain's fix-up, P0[I_b, I_b'] D_b' as before, now on the moves instead of the points --
//     wherever episodes begin and end: between episodes Y rests in LDS (`ytile`) in exactly the state the product of block row 0
//     wants (moves up to block NB - 3 applied), the moves of blocks NB - 2 and NB - 1 in `pend`; a new restart's column is
//     L^T x0 with no pending moves (column build).  A wave that has already applied block NB - 2 when the episode ends wrote
//     the state down before it did.
//   * A fragments (pair-packed: k-steps 2 vp, 2 vp + 1 in one 16-byte word): Gpack[b][beta][vp][lane] = L[16 b + (l & 15)][16 beta
//     + 4 v + (l >> 4)], Upack[b][beta][vp][lane] = L[16 b + 4 v + (l >> 4)][16 beta + (l & 15)]; requested at the top of the
//     product, ahead of the wait for the commit they do not depend on.
template <int NMW>
__device__ __attribute__((noinline)) void l2_mfma_lr_role(int m_in, int t_in) {
    constexpr int MAXC = 1;
    constexpr int KCLV = 1;
    constexpr int CSU = 1;
    constexpr int LRV = 1;
    constexpr int YU = L2_YU;
    const int tile = l2_uni(t_in);
    L2_LDS_VIEW(tile)
    (void)clsb;
    (void)fixp; (void)gtile; (void)DU2; (void)sc2; (void)cshare; (void)slk; (void)TC; (void)of0; (void)ovis; (void)oacc; (void)oswp; (void)sid; (void)snew;
    (void)sfin; (void)ost; (void)ctl; (void)cst; (void)sseed; (void)sfirst; (void)p1key; (void)p1upd; (void)p1fin; (void)p1sw; (void)p1st; (void)gatep; (void)simdof; (void)p1cols; (void)jn;
    const int lane = threadIdx.x & 63;
    const int m = l2_uni(m_in);
    const double *pG = l2_uni(par->Gpack), *pU = l2_uni(par->Upack);
    const int NB = l2_uni(par->NB), RB = l2_uni(par->RB);
    const int nu = (RB > m) ? (RB - m + NMW - 1) / NMW : 0;       // blocks of Y this wave owns (<= YU)
    const int64_t gmax = (int64_t)1 << 40;         // the role ends through RQ_STOP
    const __amdgpu_buffer_rsrc_t grsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(pG), 0, NB * RB * 2048, 0x00020000);
    const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(pU), 0, NB * RB * 2048, 0x00020000);
    const unsigned vlane = (unsigned)lane * 16u;
    typedef unsigned rq_u4 __attribute__((ext_vector_type(4)));
    #define L2_LDF(dst, rs, soff) { const rq_u4 t0_ = __builtin_amdgcn_raw_buffer_load_b128(rs, vlane, (soff), 0);           \
                                    const rq_u4 t1_ = __builtin_amdgcn_raw_buffer_load_b128(rs, vlane + 1024u, (soff), 0);   \
                                    (dst)[0] = __builtin_bit_cast(v2d_, t0_); (dst)[1] = __builtin_bit_cast(v2d_, t1_); }
    typedef __attribute__((address_space(3))) double rq_lds_d;
    typedef __attribute__((address_space(3))) const double rq_lds_cd;
    const int yoff = (lane >> 4) * 16 + (lane & 15);
    rq_lds_d *ybase = (rq_lds_d *)(ytile + yoff);
    rq_lds_cd *rbase = (rq_lds_cd *)(ring + lane);
    rq_lds_cd *pbase = (rq_lds_cd *)(pend + lane);
    asm volatile("" : "+v"(ybase), "+v"(rbase), "+v"(pbase));
    v4d_ y[YU];
    #pragma unroll
    for (int u = 0; u < YU; u++)
    #pragma unroll
        for (int v = 0; v < 4; v++) y[u][v] = (u < nu) ? ybase[(16 * (m + NMW * u) + 4 * v) * 16] : 0.0;
    auto save_y = [&]() {
    #pragma unroll
        for (int u = 0; u < YU; u++)
            if (u < nu) {
    #pragma unroll
                for (int v = 0; v < 4; v++) ybase[(16 * (m + NMW * u) + 4 * v) * 16] = y[u][v];
            }
    };
    int row = 0, spins = 0;
    int yrow = 0;                                  // Y holds the moves up to block yrow - 3 (what the product of block row yrow wants)
    int64_t iy = 0;                                // ... applied for product iy
    for (int64_t i = 0; i < gmax; i++) {
        const int row2 = (row + 1 == NB) ? 0 : row + 1;
        const int ub = row >= 3 ? row - 3 : row - 3 + NB;      // the block whose moves this product applies first
        v2d_ aU[YU][2], aG[YU][2];
    #pragma unroll
        for (int u = 0; u < YU; u++)
            if (u < nu) {      // wave-uniform
                L2_LDF(aU[u], ursrc, (ub * RB + m + NMW * u) * 2048)
                L2_LDF(aG[u], grsrc, (row * RB + m + NMW * u) * 2048)
            }
        bool stop = false;
        if (i >= 3) {
            for (;;) {
                const rq_i4 s4 = rq_sync_read(sy);
                if (s4[RQ_STOP]) { stop = true; break; }
                if (s4[RQ_COMMIT] >= (int)i - 2) break;
                if (++spins > L2_WD) { stop = true; rq_sync_write(sy, L2_ABORT, 1, lane); rq_sync_write(sy, RQ_STOP, 1, lane); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (stop) break;
            spins = 0;
        }
        if (i >= 1) {
            if (row == 1) save_y();                // the state between episodes: before the moves of block NB - 2 go in
            rq_lds_cd *src = (i >= 3) ? rbase + (int)((i - 3) & 3) * 256 : pbase + (int)(i - 1) * 256;
            double bd[4];
    #pragma unroll
            for (int q = 0; q < 4; q++) bd[q] = src[q * 64];
    #pragma unroll
            for (int v = 0; v < 4; v++)
    #pragma unroll
                for (int u = 0; u < YU; u++)
                    if (u < nu) y[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(aU[u][v >> 1][v & 1], bd[v], y[u], 0, 0, 0);
            yrow = row; iy = i;
        }
        v4d_ acc = {0.0, 0.0, 0.0, 0.0}, acc1 = acc;
    #pragma unroll
        for (int u = 0; u < YU; u++)
            if (u < nu) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aG[u][0][0], y[u][0], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aG[u][0][1], y[u][1], acc1, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aG[u][1][0], y[u][2], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aG[u][1][1], y[u][3], acc1, 0, 0, 0);
            }
        acc = acc + acc1;
        if (i >= 2) {
            for (;;) {
                const rq_i4 s4 = rq_sync_read(sy);
                if (s4[RQ_STOP]) { stop = true; break; }
                if (s4[RQ_CONS] >= (int)i - 1) break;
                if (++spins > L2_WD) { stop = true; rq_sync_write(sy, L2_ABORT, 1, lane); rq_sync_write(sy, RQ_STOP, 1, lane); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (stop) break;
            spins = 0;
        }
        {
            double *part = part2 + (int)(i & 1) * NMW * 256 + m * 256;
    #pragma unroll
            for (int v = 0; v < 4; v++) part[v * 64 + (lane & 15) * 4 + (lane >> 4)] = acc[v];
        }
        rq_sync_write(sy, RQ_PARTS + m, (int)i + 1, lane);
        row = row2;
    }
    // ---- leave Y in the state the next episode's first product wants (see the header): an episode that ended at a sweep
    // boundary finds this wave one product behind (apply block NB - 3), on it, or one ahead (already written down)
    // (a wave may even be TWO ahead: product g + 3 passes its commit wait -- block NB - 1 is committed -- and applies its moves
    //  before it stops at the slot wait; rows 1 and 2 both find the state written down at row 1)
    if (yrow == NB - 1 && iy >= 3) {
        v2d_ aU[YU][2];
    #pragma unroll
        for (int u = 0; u < YU; u++)
            if (u < nu) L2_LDF(aU[u], ursrc, ((NB - 3) * RB + m + NMW * u) * 2048)
        rq_lds_cd *src = rbase + (int)((iy - 2) & 3) * 256;
        double bd[4];
    #pragma unroll
        for (int q = 0; q < 4; q++) bd[q] = src[q * 64];
    #pragma unroll
        for (int v = 0; v < 4; v++)
    #pragma unroll
            for (int u = 0; u < YU; u++)
                if (u < nu) y[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(aU[u][v >> 1][v & 1], bd[v], y[u], 0, 0, 0);
        save_y();
    } else if (yrow == 0) {
        save_y();
    }
    #undef L2_LDF
}

// ========================================================================== chain role (a real function as well)
template <int NMW, int CS, int KIND, int TILES, int LR>
__device__ __attribute__((noinline)) void l2_chain_role(int t_in) {
    constexpr bool MULTI = KIND == L2_KIND_GENK || KIND == L2_KIND_LINK;       // several classes / two constraints per coordinate
    constexpr int BK = KIND == L2_KIND_GENK ? L2_KIND_GEN : KIND == L2_KIND_LINK ? L2_KIND_LIN : KIND;      // the arithmetic of a step
    constexpr int MAXC = MULTI ? 2 : 1;
    constexpr int KCLV = MULTI ? L2_KCL : 1;
    constexpr int CSU = CS > 0 ? CS : 1;
    constexpr int LRV = LR;                        // factored objective (l2_mfma_lr_role): the ring carries the block's MOVES, no share
    static_assert(!LR || CS == 0, "factored objective: the chain has no share of the contraction");
    static_assert(!MULTI || (TILES == 1 && !LR), "multi-class kinds: one tile per workgroup, no objective factor");
    constexpr bool PRE = BK == L2_KIND_LIN;        // a restart's phase 2 starts with a frozen sweep that evaluates f0
    const int tile = TILES > 1 ? l2_uni(t_in) : 0; // the tile whose 16 slots this chain steps
    L2_LDS_VIEW(tile)
    (void)ytile; (void)pend; (void)clsb;
    (void)slk; (void)snew; (void)p1key; (void)p1upd; (void)p1fin; (void)p1sw; (void)p1st; (void)gatep; (void)simdof; (void)p1cols; (void)jn;
    const int lane = threadIdx.x & 63, r = lane >> 2, gq = lane & 3;
    LG const double *Apk = l2_g(l2_uni(par->Apack));
    LG const double *Apk2 = l2_g(l2_uni(par->Apack2));
    LG const double *Dpk = l2_g(l2_uni(par->Dpack));
    LG const double *Spk = l2_g(l2_uni(par->Spack));
    LG double *Xg = l2_g(l2_uni(par->Xg));
    LG const int *pnext = l2_g(l2_uni(par->next));
    LG const int *clsg = l2_g(l2_uni(par->cls));
    unsigned long long *pprof = l2_uni(par->prof);
    const int pRtotal = l2_uni(par->Rtotal);
    const int NB = l2_uni(par->NB), KS = l2_uni(par->KS), nlast = l2_uni(par->nlast);
    struct { int64_t n; double r0; } P = {(int64_t)l2_uni(par->n), l2_uni(par->r0)};
    struct { double tol, fbound; int64_t num_iters; } a = {l2_uni(par->tol), l2_uni(par->fbound), (int64_t)l2_uni(par->num_iters)};
    const int64_t gmax = (int64_t)1 << 40;         // the role ends when nothing is live / at the episode's end
    // ========================================================================== chain role
    __builtin_amdgcn_s_setprio(3);
    // feasible set of this lane's restart for the episode (from the table the refill step keeps)
    const int Un = TC.n[r], Uslow = TC.slow[r];
    const double Ul0 = TC.lo[r], Uh0 = TC.hi[r], Ul1 = TC.lo[16 + r], Uh1 = TC.hi[16 + r];
    const bool two = Un >= 2;
    // BAND: [-symb, -syma] u [syma, symb], near-tie = vertex within thr of 0
    // GEN : [Ul0, Uh0] (u [Ul1, Uh1]); the vertex is projected onto the interval on its side of the gap's midpoint
    // LIN : the lowest / highest end point against the slope; near-tie = |slope| below tl
    const double thr = two ? 1e-7 * (Ul1 - Uh0) : 0.0;
    const double syma = two ? Ul1 : 0.0, symb = two ? Uh1 : Uh0;
    const double gmid = two ? 0.5 * (Uh0 + Ul1) : QM_INF;
    const double linL = Ul0, linH = two ? Uh1 : Uh0;
    // candidates of the reference's end-point comparison (utilities.py:275-288) differ by slope x distance; they
    // are told apart safely when that exceeds 1e-12 of the objective's scale (rounding: 1e-16 of it)
    auto lin_tl = [&](double l0_, double h0_, double l1_, double h1_, bool two_) {
        const double lL = l0_, lH = two_ ? h1_ : h0_;
        const double w0 = h0_ - l0_, w1 = two_ ? h1_ - l1_ : w0;
        const double wmin = w0 < w1 ? w0 : w1;
        const double hh = fabs(lL) > fabs(lH) ? fabs(lL) : fabs(lH);
        const double scale = a.fbound * (hh * hh > 1.0 ? hh * hh : 1.0);
        return (wmin > 0.0) ? 0.5e-12 * scale / wmin : QM_INF;
    };
    double tl = 0.0;
    if (BK == L2_KIND_LIN && !MULTI) tl = lin_tl(Ul0, Uh0, Ul1, Uh1, two);
    struct { int upd_counter, visits, accepted, sweeps, status; bool conv; } S;      // (32-bit in the loop: < 2^31 visits per restart)
    S.upd_counter = (int)cst[0 * 64 + lane]; S.visits = (int)cst[1 * 64 + lane]; S.accepted = (int)cst[2 * 64 + lane];
    S.sweeps = (int)cst[3 * 64 + lane]; S.conv = cst[4 * 64 + lane] != 0; S.status = (int)cst[5 * 64 + lane];
    double fpart = __longlong_as_double(cst[6 * 64 + lane]);
    // The objective of the result is not tracked from an evaluated start value: `facc` sums x_i ((P0 x)_i + q_i) over
    // the visits since the restart's last move -- when it converges (n visits without a move, qcqp.py:172-176) those
    // are all n coordinates at the FINAL point: f0(x) - r0 freshly evaluated from the products the sweep computed
    // anyway.  A restart that stops otherwise (sweep limit, gate not passed) takes one FROZEN sweep (no moves, not
    // counted) that sums the same terms.  fpart tracks the objective through the moves: relative to the start of
    // phase 2, or -- linear kind, where the reference's end-point comparison works on rounded absolute values --
    // from f0 evaluated by a frozen sweep before the first real one (`pre`).
    bool frz = (cst[7 * 64 + lane] & 1) != 0, done = (cst[7 * 64 + lane] & 2) != 0, pre = PRE && (cst[7 * 64 + lane] & 4) != 0;
    // factored objective: a restart's first sweep in its slot LOADS it -- Y = L^T x0 is what the products' own updates make of
    // "every block moves from 0 to x0": no decisions, no counters, the block's moves are the points themselves; the sweeps start
    // (or the frozen sweep of a restart that did not pass the gate) at the next boundary.  A sweep of the slot instead of a pass
    // over L per starting column (2 MB from L2 / Infinity Cache each: measured 24 % of the workgroups' time).
    bool ld = LR && (cst[7 * 64 + lane] & 8) != 0;
    double facc = __longlong_as_double(cst[8 * 64 + lane]);
    const bool occupied = sid[r] >= 0;
    const RqOwn cown = l2_own(NB, CS, -1, NMW);
    v2d_ arC[2 * CSU];
    double afix[4] = {0.0, 0.0, 0.0, 0.0}, afix2[4] = {0.0, 0.0, 0.0, 0.0};
    v4d_ carry = {0.0, 0.0, 0.0, 0.0};      // the block rewritten last times the fragments of the row after next
    double xon[4], d4[4], s3 = 0.0;         // prefetched for the next interval: x of the block, its staged operands
    {
        // staged operands of block 0, x of block 0
    #pragma unroll
        for (int e = 0; e < 4; e++) d4[e] = Dpk[lane + 64 * e];
        if (lane < 48) s3 = Spk[lane];
    #pragma unroll
        for (int v = 0; v < 4; v++) xon[v] = Xg[(4 * v + gq) * 16 + r];
    #pragma unroll
        for (int e = 0; e < 4; e++) DU2[lane + 64 * e] = d4[e];
        if (lane < 48) sc2[lane] = s3;
        if (MULTI && lane < 16) clsb[lane] = (lane < l2_uni(par->n)) ? clsg[lane] : 0;      // (a padded coordinate: class 0, see the column build)
        // virtual interval before the episode: the two blocks that a sweep rewrites last (NB - 2, NB - 1) times the
        // fragments of block rows 0 and 1, exactly as the end of a sweep leaves them (carry: NB - 1 x row 1; the
        // chain's plane: NB - 2 and NB - 1 x row 0 + the chain's share of row 0 without those two)
        v4d_ c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
        const int bl2 = NB - 2, bl1 = NB - 1;
        double f0a[4], f0b[4], f1b[4], x2[4], x1[4];
    #pragma unroll
        for (int u = 0; u < 4; u++) {
            f0a[u] = Apk[((int64_t)0 * KS + 4 * bl2 + u) * 64 + lane];
            f0b[u] = Apk[((int64_t)0 * KS + 4 * bl1 + u) * 64 + lane];
            f1b[u] = Apk[((int64_t)1 * KS + 4 * bl1 + u) * 64 + lane];
            // (factored objective: the products carry Y with the moves up to block NB - 3; the chain supplies the MOVES of the two
            //  blocks after it -- zero for a restart that starts here)
            x2[u] = LR ? pend[u * 64 + lane] : Xg[(4 * bl2 + u) * 64 + lane];
            x1[u] = LR ? pend[256 + u * 64 + lane] : Xg[(4 * bl1 + u) * 64 + lane];
        }
    #pragma unroll
        for (int u = 0; u < 4; u++) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0a[u], x2[u], c0, 0, 0, 0);
    #pragma unroll
        for (int u = 0; u < 4; u++) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0b[u], x1[u], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1b[u], x1[u], c1, 0, 0, 0);
        }
        carry = c1;
        if (CS > 0) {
            l2_load_A<CSU>(arC, Apk2, KS, cown, lane, 0);
            double bqC[4 * CSU];
    #pragma unroll
            for (int U = 0; U < CSU; U++) {
                const int bb = rq_block(cown, U);
    #pragma unroll
                for (int q = 0; q < 4; q++) bqC[4 * U + q] = Xg[(4 * bb + q) * 64 + lane];
            }
    #pragma unroll
            for (int U = 0; U < CSU; U++)
    #pragma unroll
                for (int q = 0; q < 4; q++) cshare[U * 256 + q * 64 + lane] = bqC[4 * U + q];
            c0 = l2_product<CSU>(arC, bqC, Apk2, KS, cown, lane, rq_slot(cown, bl1), rq_slot(cown, bl2), 1, c0);
        }
    #pragma unroll
        for (int v = 0; v < 4; v++) fixp[v * 64 + (lane & 15) * 4 + (lane >> 4)] = c0[v];
    }
    const double tolv = a.tol;
    int b = 0, spins = 0;
    bool wdog = false;
    const bool prof_on = pprof != nullptr;
    long long pw_part = 0, pt_sum = 0, pt_steps = 0, pt_end = 0, pt_fix = 0, pt_req = 0, ptl = 0;
    if (prof_on) { ptl = (long long)__builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(pprof + 18, (unsigned long long)(ptl - *(long long *)(ctl + 6))); }
    int pn_int = 0, pn_gen = 0;
#define L2_TICK(acc) if (prof_on) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); acc += now_ - ptl; ptl = now_; }
    int64_t last_g = 0;
    for (int64_t g = 0; g < gmax; g++) {
        last_g = g;
        const int bn = (b + 1 == NB) ? 0 : b + 1, bn2 = (bn + 1 == NB) ? 0 : bn + 1;
        const int bprev = (b == 0) ? NB - 1 : b - 1;
        const int cur = (int)(g & 1);
        const double *DU = DU2 + cur * 256, *hqb = sc2 + cur * 48, *rtb = sc2 + cur * 48 + 16, *dgb = sc2 + cur * 48 + 32;
        const double *part = part2 + cur * NMW * 256;
        double *rb = ring + (int)(g & 3) * 256;
        const int ncol = (b == NB - 1) ? nlast : 16;         // real coordinates of the block (the padded ones never move)
        // ---- requests whose answers the NEXT interval needs: staged operands and x of block bn (committed NB - 1
        // intervals ago, or part of the tile the episode started from)
        double xo[4];
    #pragma unroll
        for (int v = 0; v < 4; v++) xo[v] = xon[v];
    #pragma unroll
        for (int e = 0; e < 4; e++) d4[e] = Dpk[(int64_t)bn * 256 + lane + 64 * e];
        if (lane < 48) s3 = Spk[(int64_t)bn * 48 + lane];
        int c3 = 0;
        if (MULTI && lane < 16) c3 = (16 * bn + lane < (int)P.n) ? clsg[16 * bn + lane] : 0;
    #pragma unroll
        for (int v = 0; v < 4; v++) xon[v] = Xg[(16 * (int64_t)bn + 4 * v + gq) * 16 + r];
        {   // A fragments of this block's k-steps in the next two block rows (the chain's contribution to both)
            LG const double *ap = Apk + ((int64_t)bn * KS + 4 * b) * 64 + lane;
            LG const double *ap2 = Apk + ((int64_t)bn2 * KS + 4 * b) * 64 + lane;
    #pragma unroll
            for (int u = 0; u < 4; u++) { afix[u] = ap[u * 64]; afix2[u] = ap2[u * 64]; }
        }
        // ---- partial tiles of product g
        const long long tw0 = prof_on ? (long long)__builtin_amdgcn_s_memtime() : 0;
        for (;;) {
            const rq_i4 p4 = rq_sync_read(sy + RQ_PARTS);
            int lo4 = p4[0] < p4[1] ? p4[0] : p4[1];
            const int lo2 = p4[2] < p4[3] ? p4[2] : p4[3];
            lo4 = lo4 < lo2 ? lo4 : lo2;
            if (NMW > 4) {
                const rq_i4 q4 = rq_sync_read(sy + RQ_PARTS + 4);
                const int l1 = q4[0] < q4[1] ? q4[0] : q4[1], l2 = q4[2] < q4[3] ? q4[2] : q4[3];
                lo4 = lo4 < l1 ? lo4 : l1;
                lo4 = lo4 < l2 ? lo4 : l2;
            }
            if (lo4 >= (int)g + 1) break;
            if (++spins > L2_WD) { wdog = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (wdog) { rq_sync_write(sy, L2_ABORT, 1, lane); break; }
        spins = 0;
        if (prof_on) { pt_req += tw0 - ptl; ptl = (long long)__builtin_amdgcn_s_memtime(); pw_part += ptl - tw0; pn_int++; }
        // ---- G + q/2 of the lane's own columns: its own plane, then the partial tiles, in a fixed order, then q/2
        double gb[4], xn[4], rto[4], t2o[4], hq4[4];
    #pragma unroll
        for (int v = 0; v < 4; v++) {
            double s = fixp[v * 64 + lane];
    #pragma unroll
            for (int w = 0; w < NMW; w++) s += part[w * 256 + v * 64 + lane];
            hq4[v] = hqb[4 * v + gq];
            s += hq4[v];
            gb[v] = s;
            gtile[(4 * v + gq) * 16 + r] = s;             // kept for the generic path (the partial tiles are released now)
        }
        rq_sync_write(sy, RQ_CONS, (int)g + 1, lane);     // the partial tiles have been read (LDS is in order per wave)
    #pragma unroll
        for (int v = 0; v < 4; v++) { rto[v] = rtb[4 * v + gq]; t2o[v] = dgb[4 * v + gq]; }
        if (b == 0 && !S.conv && S.sweeps >= (int)a.num_iters) {      // sweep limit reached (qcqp.py:160): the restart is done
            S.conv = true;
            frz = true; facc = 0.0;                               // ... after one frozen sweep that evaluates its objective
        }
        if (b == 0 && !S.conv) S.sweeps++;
        const bool act = !S.conv;
        // the feasible set of the lane's restart for each of its four columns: ONE set for the single-class kinds (registers of the
        // episode), the set of the column's class from the slots' table for the multi-class kinds
        int nv[4], slv[4];
        double l0v[4], h0v[4], l1v[4], h1v[4], midv[4], thrv[4], tlv[4], tolev[4];
    #pragma unroll
        for (int v = 0; v < 4; v++) {
            if (MULTI) {
                const int s_ = clsb[cur * 16 + 4 * v + gq] * 16 + r;
                nv[v] = TC.n[s_]; slv[v] = TC.slow[s_];
                l0v[v] = TC.lo[s_]; h0v[v] = TC.hi[s_]; l1v[v] = TC.lo[16 * KCLV + s_]; h1v[v] = TC.hi[16 * KCLV + s_];
                const bool two_ = nv[v] >= 2;
                midv[v] = two_ ? 0.5 * (h0v[v] + l1v[v]) : QM_INF;
                thrv[v] = two_ ? 1e-7 * (l1v[v] - h0v[v]) : 0.0;
                tlv[v] = (BK == L2_KIND_LIN) ? lin_tl(l0v[v], h0v[v], l1v[v], h1v[v], two_) : 0.0;
                if (BK == L2_KIND_LIN) h1v[v] = two_ ? h1v[v] : h0v[v];          // (linear kind: h1v = the highest end point)
            } else {
                nv[v] = Un; slv[v] = Uslow; l0v[v] = Ul0; h0v[v] = Uh0; l1v[v] = Ul1; h1v[v] = (BK == L2_KIND_LIN) ? linH : Uh1;
                midv[v] = gmid; thrv[v] = thr; tlv[v] = tl;
            }
            tolev[v] = (act && nv[v] > 0) ? tolv : QM_INF;     // a restart that is not sweeping never moves (the padded coordinates of
                                                               // the last block hold a fixed point of the step: see the column build)
        }
        L2_TICK(pt_sum)
        // ---- the 16 steps: only what the next step waits for
    #pragma unroll
        for (int c = 0; c < 16; c++) {
            const int v = c >> 2, go = c & 3;
            // every lane works on its own column 4 v + gq; only the owner quad-lane (gq == go) is at step c
            double pick;
            if (BK == L2_KIND_BAND) {
                const double xv = __builtin_fma(-gb[v], rto[v], xo[v]);          // vertex of the scalar objective
                pick = __builtin_copysign(fmin(fmax(fabs(xv), syma), symb), xv);
            } else if (BK == L2_KIND_GEN) {
                const double xv = __builtin_fma(-gb[v], rto[v], xo[v]);
                const double p0 = fmin(fmax(xv, l0v[v]), h0v[v]), p1 = fmin(fmax(xv, l1v[v]), h1v[v]);
                pick = (xv > midv[v]) ? p1 : p0;
            } else {
                pick = (gb[v] > 0.0) ? l0v[v] : h1v[v];                           // linear: the end point against the slope
            }
            const double dlt = pick - xo[v];
            const double dl = (fabs(dlt) > tolev[v]) ? dlt : 0.0;
            double delta;
            if (go == 0) delta = rq_quad_bcast<0x00>(dl);
            else if (go == 1) delta = rq_quad_bcast<0x55>(dl);
            else if (go == 2) delta = rq_quad_bcast<0xAA>(dl);
            else delta = rq_quad_bcast<0xFF>(dl);
    #pragma unroll
            for (int v2 = v; v2 < 4; v2++) gb[v2] = __builtin_fma(DU[c * 16 + 4 * v2 + gq], delta, gb[v2]);
        }
        L2_TICK(pt_steps)
        // ---- once per block, per own column: the decision again from the frozen G (bit-identical to what the step
        // computed when the lane was the owner), new x, near-tie test, move mask, objective tracking
        bool needgen = false;      // a real column of this lane whose decision is near a tie or whose set has an infinite end / a third interval
        unsigned mv = 0;
        double fadd = 0.0;
    #pragma unroll
        for (int v = 0; v < 4; v++) {
            double pick;
            bool far;
            if (BK == L2_KIND_BAND) {
                const double xv = __builtin_fma(-gb[v], rto[v], xo[v]);
                pick = __builtin_copysign(fmin(fmax(fabs(xv), syma), symb), xv);
                far = fabs(xv) > thr;                                             // false for NaN as well
            } else if (BK == L2_KIND_GEN) {
                const double xv = __builtin_fma(-gb[v], rto[v], xo[v]);
                const double p0 = fmin(fmax(xv, l0v[v]), h0v[v]), p1 = fmin(fmax(xv, l1v[v]), h1v[v]);
                pick = (xv > midv[v]) ? p1 : p0;
                far = fabs(xv - midv[v]) > thrv[v];
            } else {
                pick = (gb[v] > 0.0) ? l0v[v] : h1v[v];
                far = fabs(gb[v]) > tlv[v];
            }
            const double dlt = pick - xo[v];
            const bool mvd = fabs(dlt) > tolev[v];
            const double d = mvd ? dlt : 0.0;
            xn[v] = mvd ? pick : xo[v];
            needgen = needgen || (nv[v] > 0 && (4 * v + gq < ncol ? (!far || slv[v] != 0) : (!MULTI && slv[v] != 0)));
            mv |= mvd ? (1u << (4 * v + gq)) : 0u;
            // f(x + d e_i) - f(x) = d (2 (P x)_i + q_i + P_ii d) = d (t2 d + 2 g):  g = G_i + q_i / 2 contains P_ii x_i
            fadd = __builtin_fma(d, __builtin_fma(t2o[v], d, gb[v] + gb[v]), fadd);
        }
        mv = rq_quad_or(mv);                                                       // bit c = coordinate c moved
        // per RESTART: does the block need the reference's arithmetic?  Only those restarts walk the generic loop
        const bool redo = rq_quad_or((act && needgen) ? 1u : 0u) != 0u;
        auto fast_commit = [&]() {
            if (act) {
                fpart += fadd;
                const int accn = __builtin_popcount(mv);
                const int upd = mv ? (ncol - 1 - (31 - __builtin_clz(mv))) : S.upd_counter + ncol;
                S.accepted += accn;
                const int over = upd - (int)P.n;
                S.visits += ncol - (over > 0 ? over : 0);
                S.upd_counter = upd;
                if (over >= 0) S.conv = true;
                // the window: visits after the block's last move (all of them if none moved), up to the visit that
                // completes the n consecutive visits without a move
                const int cl = mv ? 31 - __builtin_clz(mv) : -1, ce = ncol - 1 - (over > 0 ? over : 0);
                double w = 0.0;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int c = 4 * v + gq;
                    w = (c > cl && c <= ce) ? __builtin_fma(xn[v], gb[v] + hq4[v], w) : w;
                }
                facc = (cl >= 0 ? 0.0 : facc) + w;
                if (over >= 0) done = true;
            } else if ((frz || pre) && !ld) {
                double w = 0.0;
#pragma unroll
                for (int v = 0; v < 4; v++) w = __builtin_fma(xo[v], gb[v] + hq4[v], w);
                facc += w;
            }
#pragma unroll
            for (int v = 0; v < 4; v++) rb[(4 * v + gq) * 16 + r] = LR ? (ld ? xo[v] : xn[v] - xo[v]) : xn[v];
        };
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(redo) == 0ull, 1)) {
            fast_commit();
        } else {
            // ---- generic loop (rare): the reference's arithmetic on the G tile kept from the block's start and the block's
            // x in its ring slot; all four lanes of a quad walk their restart redundantly (same values, benign identical
            // LDS writes)
            if (prof_on) pn_gen++;
            if (!redo) fast_commit();
            else {
                ChainState G;
                G.fcur = rq_quad_sum(fpart); G.upd_counter = S.upd_counter; G.visits = S.visits; G.accepted = S.accepted;
                G.sweeps = S.sweeps; G.conv = S.conv; G.status = S.status;
                double fa = rq_quad_sum(facc);
                const uint64_t dseed = sseed[r], dfirst = sfirst[r];
                FeasSet<MAXC> C;
                C.n = Un;
#pragma unroll
                for (int j = 0; j <= MAXC; j++) { C.lo[j] = TC.lo[j * 16 * KCLV + r]; C.hi[j] = TC.hi[j * 16 * KCLV + r]; }
                int cslow = Uslow;
                double ctl_ = tl, clinL = linL, clinH = linH;
#pragma unroll
                for (int v = 0; v < 4; v++) rb[(4 * v + gq) * 16 + r] = xo[v];
                for (int c = 0; c < ncol; c++) {
                    if (MULTI) {       // the set of the coordinate's class
                        const int s_ = clsb[cur * 16 + c] * 16 + r;
                        C.n = TC.n[s_]; cslow = TC.slow[s_];
#pragma unroll
                        for (int j = 0; j <= MAXC; j++) { C.lo[j] = TC.lo[j * 16 * KCLV + s_]; C.hi[j] = TC.hi[j * 16 * KCLV + s_]; }
                        if (BK == L2_KIND_LIN) {
                            const bool two_ = C.n >= 2;
                            clinL = C.lo[0]; clinH = two_ ? C.hi[1] : C.hi[0];
                            ctl_ = lin_tl(C.lo[0], C.hi[0], C.lo[1], C.hi[1], two_);
                        }
                    }
                    const int64_t i = 16 * (int64_t)b + c;
                    const double t2g = dgb[c];
                    const double xi = rb[c * 16 + r];
                    const double hq = hqb[c];
                    const double gc = gtile[c * 16 + r];
                    const double t1 = 2.0 * ((gc - hq) - t2g * xi) + (hq + hq);
                    const double t0 = G.fcur - xi * (t2g * xi + t1);
                    DrawKey dk{dseed, dfirst + (uint64_t)sid[r], (uint32_t)i, (uint32_t)(G.sweeps - 1) | 0x80000000u, 0u};
                    double xnew = xi;
                    int got;
                    if (BK == L2_KIND_LIN && cslow == 0 && C.n > 0 && fabs(gc) > ctl_) {
                        // linear kind, the slope is far from a tie: the end point against the slope IS what the reference's end-point
                        // comparison returns (utilities.py:275-288 picks among the table's own values: nothing to round) -- only the
                        // coordinates that are near a tie pay for the replay.  (Unweighted MAXCUT has a tie in almost every block.)
                        xnew = (gc > 0.0) ? clinL : clinH;
                        got = G.conv ? 0 : 1;
                    } else {
                        got = G.conv ? 0 : onevar_minimise<MAXC>(t2g, t1, t0, C, dk, &xnew);
                    }
                    bool moved;
                    double delta;
                    const bool wasconv = G.conv;
                    chain_commit<MAXC>(G, got, xnew, xi, t2g, t1, t0, a.tol, P.n, moved, delta);
                    if (!wasconv) fa = moved ? 0.0 : fa + xi * (gc + hq);
                    if (moved) {
                        rb[c * 16 + r] = xnew;
                        for (int c2 = c + 1; c2 < 16; c2++) gtile[c2 * 16 + r] += DU[c * 16 + c2] * delta;
                    }
                }
                S.upd_counter = (int)G.upd_counter; S.visits = (int)G.visits; S.accepted = (int)G.accepted; S.conv = G.conv; S.status = G.status;
                fpart = (gq == 0) ? G.fcur : 0.0;
                facc = (gq == 0) ? fa : 0.0;
                if (S.conv) done = true;
#pragma unroll
                for (int v = 0; v < 4; v++) xn[v] = rb[(4 * v + gq) * 16 + r];
                if (LR) {
#pragma unroll
                    for (int v = 0; v < 4; v++) rb[(4 * v + gq) * 16 + r] = xn[v] - xo[v];
                }
            }
        }
        // the block goes to the global tile as well (the next sweep's prefetch, the next episode's operands, the result)
    #pragma unroll
        for (int v = 0; v < 4; v++) Xg[(16 * (int64_t)b + 4 * v + gq) * 16 + r] = xn[v];
        rq_sync_write(sy, RQ_COMMIT, (int)g + 1, lane);   // block b is in the ring
        L2_TICK(pt_end)
        if (b == NB - 1) {
            if (ld) { ld = false; if (!frz) S.conv = false; }      // loaded: the restart starts sweeping (or takes its frozen sweep) with the next block
            else if (frz) { frz = false; done = true; }   // the frozen sweep is complete
            if (PRE && pre) {
                // f0 at the start of phase 2: the restart starts sweeping with the next block
                const double f0s = rq_quad_sum(facc) + P.r0;
                fpart = (gq == 0) ? f0s : 0.0;
                facc = 0.0; pre = false; S.conv = false;
            }
        }
        const unsigned long long livem = __builtin_amdgcn_ballot_w64(!S.conv || frz || pre || ld);
        if (TILES == 1) {
            if (livem == 0ull) break;
            if (b == NB - 1) {
                // sweep boundary: slots whose restart is done can take a new restart -- end the episode if the queue has one
                const bool fin = !occupied || done;
                const unsigned long long finm = __builtin_amdgcn_ballot_w64(fin);
                if (finm == ~0ull) break;
                if (finm != 0ull) {
                    const bool more = l2_load_int(pnext) < pRtotal;
                    if (more) break;
                }
            }
        } else if (b == NB - 1) {
            // Two tiles per workgroup: the episode ends for BOTH chains at the same sweep boundary (the build and the write-out
            // are the workgroup's).  Once per sweep the chain of tile 1 posts what it sees -- something live, all slots finished,
            // some slot finished -- and waits for the verdict; the chain of tile 0 adds its own, reads the queue ONCE (two reads
            // could disagree) and posts the verdict.  Words carry the sweep number of the episode: nothing to reset in between.
            // (A tile with nothing live waits for the boundary instead of leaving mid-sweep: its slots are idle either way.)
            const unsigned long long finm = __builtin_amdgcn_ballot_w64(!occupied || done);
            const int mine = (livem != 0ull ? 1 : 0) | (finm == ~0ull ? 2 : 0) | (finm != 0ull ? 4 : 0);
            const int tag = ((int)(g / NB) + 1) << 8;
            rq_lds_int *jw = (rq_lds_int *)jn;
            int verdict = 0;
            if (tile == 1) rq_sync_write(jw, 1, tag | mine, lane);
            for (;;) {
                int w = *(volatile rq_lds_int *)(jw + (tile == 1 ? 0 : 1));
                asm volatile("" ::: "memory");
                w = __builtin_amdgcn_readfirstlane(w);
                if ((w & ~0xff) == tag) { verdict = w & 0xff; break; }
                if (++spins > L2_WD) { wdog = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (wdog) { rq_sync_write(sy, L2_ABORT, 1, lane); break; }
            spins = 0;
            if (tile == 0) {
                const int both = mine | verdict, all = mine & verdict;
                bool stop = !(both & 1) || (all & 2);
                if (!stop && (both & 4)) stop = l2_load_int(pnext) < pRtotal;
                rq_sync_write(jw, 0, tag | (stop ? 1 : 0), lane);
                verdict = stop ? 1 : 0;
            }
            if (verdict & 1) break;
        }
        // ---- the chain's part of the next products: the block just committed times the fragments of the next TWO block
        // rows, and its own share of the next row
        {
            v4d_ acc = carry, acc2 = {0.0, 0.0, 0.0, 0.0};
            double xb4[4];
    #pragma unroll
            for (int u = 0; u < 4; u++) xb4[u] = rb[u * 64 + lane];
    #pragma unroll
            for (int u = 0; u < 4; u++) {
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(afix[u], xb4[u], acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(afix2[u], xb4[u], acc2, 0, 0, 0);
            }
            carry = acc2;
            if (CS > 0) {
                const int us = rq_slot(cown, b);
                if (us >= 0) {                       // wave-uniform: the block just committed belongs to the chain's share
    #pragma unroll
                    for (int q = 0; q < 4; q++) cshare[us * 256 + q * 64 + lane] = xb4[q];
                }
                double bqC[4 * CSU];                 // (short-lived: the step loop's registers are free here)
    #pragma unroll
                for (int U = 0; U < CSU; U++)
    #pragma unroll
                    for (int q = 0; q < 4; q++) bqC[4 * U + q] = cshare[U * 256 + q * 64 + lane];
                acc = l2_product<CSU>(arC, bqC, Apk2, KS, cown, lane, us, rq_slot(cown, bprev), bn2, acc);
            }
    #pragma unroll
            for (int v = 0; v < 4; v++) fixp[v * 64 + (lane & 15) * 4 + (lane >> 4)] = acc[v];
        }
        // ---- staged operands of the next block into the other parity's buffers
        {
            const int nx = (int)((g + 1) & 1);
    #pragma unroll
            for (int e = 0; e < 4; e++) DU2[nx * 256 + lane + 64 * e] = d4[e];
            if (lane < 48) sc2[nx * 48 + lane] = s3;
            if (MULTI && lane < 16) clsb[nx * 16 + lane] = c3;
        }
        L2_TICK(pt_fix)
        b = bn;
    }
#undef L2_TICK
    if (LR) {
        // the moves of the two blocks committed last (blocks NB - 2 and NB - 1 when the episode ended at a sweep boundary; nothing
        // that continues otherwise) stay for the next episode: its products apply them, its chain starts from them
        const int g1 = (int)((last_g - 1) & 3), g0 = (int)(last_g & 3);
        double p1v[4], p0v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { p1v[u] = ring[g1 * 256 + u * 64 + lane]; p0v[u] = ring[g0 * 256 + u * 64 + lane]; }
#pragma unroll
        for (int u = 0; u < 4; u++) { pend[u * 64 + lane] = p1v[u]; pend[256 + u * 64 + lane] = p0v[u]; }
    }
    rq_sync_write(sy, RQ_STOP, 1, lane);
    if (prof_on && lane == 0) {
        atomicAdd(pprof + 13, (unsigned long long)pt_sum);
        atomicAdd(pprof + 14, (unsigned long long)pt_steps);
        atomicAdd(pprof + 15, (unsigned long long)pt_end);
        atomicAdd(pprof + 19, (unsigned long long)pt_req);        // issuing the interval's requests (top of the loop)
        *(long long *)(ctl + 8) = (long long)__builtin_amdgcn_s_memtime();
        atomicAdd(pprof + 7, (unsigned long long)pt_fix);
        atomicAdd(pprof + 8, (unsigned long long)pw_part);
        atomicAdd(pprof + 11, (unsigned long long)pn_int);
        atomicAdd(pprof + 12, (unsigned long long)pn_gen);
    }
    {
        cst[0 * 64 + lane] = S.upd_counter; cst[1 * 64 + lane] = S.visits; cst[2 * 64 + lane] = S.accepted;
        cst[3 * 64 + lane] = S.sweeps; cst[4 * 64 + lane] = S.conv ? 1 : 0; cst[5 * 64 + lane] = S.status;
        cst[6 * 64 + lane] = __double_as_longlong(fpart);
        cst[7 * 64 + lane] = (frz ? 1 : 0) | (done ? 2 : 0) | (pre ? 4 : 0) | (ld ? 8 : 0);
        cst[8 * 64 + lane] = __double_as_longlong(facc);
        const double ftot = rq_quad_sum(facc) + P.r0;
        const bool fin = occupied && done;
        if (gq == 0) {
            sfin[r] = fin ? 1 : 0;
            if (fin) { ovis[r] = S.visits; oacc[r] = S.accepted; oswp[r] = S.sweeps; ost[r] = S.status; of0[r] = ftot; }
        }
    }
    __builtin_amdgcn_s_setprio(0);
}

// NMW: multiplying waves per tile (3: a chain wave + three multiplying waves per tile; 7: eight waves, one per CU, 1024 < n <= 2304
//      -- the eighth wave multiplies too, beside the chain on SIMD 0)
// CS : blocks of the contraction the chain wave multiplies itself
// KIND: L2_KIND_BAND / GEN / LIN (cd_life.h)
// TILES: tiles of 16 slots per workgroup.  1: four-wave workgroups, two per CU (NMW = 3), or one eight-wave workgroup (NMW = 7).
//      2 (round 6, NMW = 3 only): ONE eight-wave workgroup per CU runs two tiles -- both chains on SIMD 0, a product stream per
//      tile on each of SIMDs 1-3, exactly what two TILES = 1 workgroups on a CU do while both are in their roles -- but the
//      episode is the workgroup's: the columns of both tiles are built by all 512 threads on a CU whose matrix pipes are idle
//      (the TILES = 1 build runs under the neighbour's product streams: 18 % of a workgroup's time against 10 %), and no tile
//      ever runs alone at the pace of a lone chain.  Per restart nothing changes: same roles, same arithmetic, same association.
template <int NMW, int CS, int KIND, int TILES, int LR>
__global__ __launch_bounds__((NMW == 3 && TILES == 1) ? 256 : 512, 2) void cd_life_kernel(CdLife2Args a0) {
    const CdLife2Args &a = a0;
    constexpr bool MULTI = KIND == L2_KIND_GENK || KIND == L2_KIND_LINK;       // several constraint classes / two constraints per coordinate
    constexpr int BK = KIND == L2_KIND_GENK ? L2_KIND_GEN : KIND == L2_KIND_LINK ? L2_KIND_LIN : KIND;
    constexpr int MAXC = MULTI ? 2 : 1;
    constexpr int KCLV = MULTI ? L2_KCL : 1;
    constexpr int CSU = CS > 0 ? CS : 1;
    constexpr int LRV = LR;                        // factored objective P0 = L L^T (l2_mfma_lr_role)
    static_assert(!MULTI || (TILES == 1 && !LR), "multi-class kinds: one tile per workgroup, no objective factor");
    constexpr int NT = (NMW == 3 && TILES == 1) ? 256 : 512;
    constexpr int NW = NT / 64;
    constexpr int NS = 16 * TILES;                 // slots of the workgroup
    constexpr int TD = l2_tile_doubles(NMW, CSU, LR);  // doubles of one tile's LDS block
    static_assert(!LR || (NMW == 3 && CS == 0 && BK != L2_KIND_LIN), "factored objective: three multiplying waves per tile, no chain share, a positive diagonal");
    constexpr bool PRE = BK == L2_KIND_LIN;        // a restart's phase 2 starts with a frozen sweep that evaluates f0 (see the chain role)
    static_assert(TILES == 1 || NMW == 3, "two tiles per workgroup: eight waves = two chains + two x three multiplying waves");
    const DevProblem &P = a.P;
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int64_t n16 = P.n16;
    const int NB = (int)P.NB, KS = (int)P.KS;
    L2_LDS_VIEW(0)                                  // tile 0's block; tile t: l2_tl(array, t, TD)
    (void)fixp; (void)gtile; (void)DU2; (void)sc2; (void)ring; (void)cshare; (void)of0; (void)ovis; (void)oacc; (void)oswp; (void)ost; (void)ytile; (void)pend; (void)clsb;
    // the feasible sets of slot `sc_` of tile `st_` at slack `slack_`: one per class (the single class of the other kinds); a set with a
    // third interval (two constraints per coordinate can leave one) takes the generic path like one with an infinite end.
    // (a macro, not a lambda: with the sets computed inside a closure the backend stopped at "illegal VGPR to SGPR copy")
#define L2_STORE_SETS(st_, sc_, slack_) {                                                                                              \
        SetTable<MAXC> T2 = TC;                                                                                                          \
        T2.lo = l2_tl(TC.lo, st_, TD); T2.hi = l2_tl(TC.hi, st_, TD); T2.n = l2_tl(TC.n, st_, TD); T2.slow = l2_tl(TC.slow, st_, TD);    \
        const int nk_ = MULTI ? a0.nclass : 1;                                                                                           \
        for (int k_ = 0; k_ < nk_; k_++) {                                                                                               \
            FeasSet<MAXC> C_;                                                                                                            \
            compute_set<MAXC>(P, P.krep[k_], slack_, C_);                                                                                \
            store_set<MAXC>(T2, 16 * k_ + (sc_), C_);                                                                                    \
            if (MULTI && C_.n > 2) T2.slow[16 * k_ + (sc_)] = 1;                                                                         \
        }                                                                                                                                \
    }
    LG double *Xg0 = l2_g(a0.scratch) + (int64_t)blockIdx.x * TILES * n16 * 16;       // this workgroup's X tiles [tile][j][16]

    // ---- roles by hardware SIMD.  The dispatcher deals the waves of a workgroup round robin over the four SIMDs starting
    // wherever the CU's pointer stands (measured, tools/ubench/ubench5.hip: wave w of a four-wave workgroup is NOT on SIMD w).
    // Both chains of a CU go to SIMD 0 (two latency-bound waves interleave well), the product streams to SIMDs 1-3.
    // If the waves do not cover the SIMDs evenly the roles fall back to the wave index (slower, equally correct).
    if ((tid0 & 63) == 0) simdof[wave] = (int)(__builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11)));       // HW_ID[5:4]
    if (NMW == 3 && TILES == 1 && tid0 == 0) {
        // which of its CU's workgroups is this one?  (round 6 experiment, off by default -- capi.hip: with the factored objective a
        // multiplying wave has 48 matrix instructions per block interval and the SIMD that carries BOTH chains of the CU sets the pace;
        // a chain beside a product stream turned out slower still)
        int arrival = 0;
        if (a0.cuslot) {
            const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
            // key: XCC, then HW_ID's SE_ID[15:13] and SH_ID[12] (four bits), then CU_ID[11:8]: one counter per compute unit
            arrival = l2_add(l2_g(a0.cuslot) + (((xcc & 15) << 8) | (((hw >> 12) & 15) << 4) | ((hw >> 8) & 15)), 1);
        }
        simdof[8] = (arrival & 1) ? (a0.rotmode == 2 ? 16 + 1 : 2) : 0;      // >= 16: a turn among the multiplying SIMDs only
    }
    if (tid0 < 64 * TILES) {
        long long *cs_ = l2_tl(cst, tid0 >> 6, TD);
#pragma unroll
        for (int f = 0; f < 10; f++) cs_[f * 64 + (tid0 & 63)] = (f == 4) ? 1 : 0;
    }
    if (tid0 < NS) { l2_tl(sid, tid0 >> 4, TD)[tid0 & 15] = -1; l2_tl(sfin, tid0 >> 4, TD)[tid0 & 15] = 0; }
    LG const CdLife *lf0 = l2_g(a0.life);
    const long long life_t0 = (tid0 == 0 && lf0->prof) ? (long long)__builtin_amdgcn_s_memtime() : 0;
    __syncthreads();
    int role, rtile = 0;                           // role: 0 chain, 1 .. NMW multiplying wave role - 1; rtile: the tile it works for
    {
        int cnt[4] = {0, 0, 0, 0};
        int rank = 0;
        const int mys = simdof[wave];
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const int s = simdof[w];
#pragma unroll
            for (int q = 0; q < 4; q++) cnt[q] += (s == q) ? 1 : 0;
            rank += (s == mys && w < wave) ? 1 : 0;
        }
        const bool even = cnt[0] == NW / 4 && cnt[1] == NW / 4 && cnt[2] == NW / 4 && cnt[3] == NW / 4;
        if (NMW == 3 && TILES == 1) {
            const int rot = simdof[8];
            if (!even) role = wave;
            else if (rot >= 16) role = mys == 0 ? 0 : 1 + ((mys - 1 + (rot - 16)) % 3);
            else role = (mys + rot) & 3;
        }
        else if (TILES == 2) { role = even ? mys : (wave & 3); rtile = even ? rank : (wave >> 2); }      // SIMD s: the waves of role s of both tiles
        else if (even) role = (mys == 0) ? (rank == 0 ? 0 : 7) : mys + 3 * rank;      // SIMD s: waves s (, s + 3); SIMD 0: the chain and wave 7
        else role = wave;
        role = __builtin_amdgcn_readfirstlane(role);
        rtile = __builtin_amdgcn_readfirstlane(rtile);
        if (tid0 == 0 && lf0->prof) {                // debug: workgroups whose waves covered the SIMDs evenly / the CU they sat on
            if (even) atomicAdd((unsigned long long *)lf0->prof + 6, 1ull);
            const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
            Xg0[0] = (double)(((xcc & 15) << 8) | (((hw >> 13) & 7) << 4) | ((hw >> 8) & 15));   // overwritten by the first column build
        }
    }
    const int nlast = (int)(P.n - 16 * (int64_t)(NB - 1));      // real coordinates of the last block (1..16)
    if (tid0 < TILES) {
        L2Par *pr = l2_tl(par, tid0, TD);
        pr->Apack = P.Apack; pr->Apack2 = P.Apack2; pr->Dpack = a0.Dpack; pr->Spack = a0.Spack;
        pr->Gpack = a0.Gpack; pr->Upack = a0.Upack; pr->RB = a0.RB; pr->pad_ = 0;
        pr->Xg = a0.scratch + ((int64_t)blockIdx.x * TILES + tid0) * n16 * 16; pr->next = a0.b.next; pr->prof = (unsigned long long *)lf0->prof;
        pr->num_iters = a.num_iters; pr->n16 = n16; pr->tol = a.tol; pr->r0 = P.r0; pr->fbound = a.fbound;
        pr->NB = NB; pr->KS = KS; pr->n = (int)P.n; pr->nlast = nlast; pr->Rtotal = (int)lf0->Rtotal; pr->dbg = a0.dbg;
        pr->cls = P.cls;
    }
    __syncthreads();

    for (;;) {
        // the lane index is made opaque per episode (otherwise every lane-dependent address of every role is hoisted out of
        // the episode loop and stays live through all roles)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const CdLife *lifep = a0.life;
        asm volatile("" : "+s"(lifep));
        const int lane = tid & 63, r = lane >> 2;
        // ================================================================ refill: free slots take the next restarts
        if (tid == 0) { ctl[0] = 0; jn[0] = 0; jn[1] = 0; }
        __syncthreads();
        if (tid < NS) {
            const int st = tid >> 4, sc = tid & 15;          // slot sc of tile st
            int id = l2_tl(sid, st, TD)[sc], nw = 0;
            if (id < 0) {
                LG const CdLife *lf = l2_g(lifep);
                const int idx = l2_add(l2_g(a0.b.next), 1);
                if (idx < (int)lf->Rtotal) {
                    id = idx; nw = 1;
                    const uint64_t pop = (uint64_t)idx / (uint64_t)lf->Rpop, rho = (uint64_t)idx % (uint64_t)lf->Rpop;
                    l2_tl(sseed, st, TD)[sc] = lf->seed + pop * lf->seed_stride;
                    l2_tl(sfirst, st, TD)[sc] = lf->first_index + pop * lf->first_stride + rho - (uint64_t)idx;    // + id = the global restart index
                }
            }
            l2_tl(sid, st, TD)[sc] = id; l2_tl(snew, st, TD)[sc] = nw;
            if (nw) { l2_tl(p1fin, st, TD)[sc] = 0; l2_tl(p1sw, st, TD)[sc] = 0; l2_tl(p1st, st, TD)[sc] = 0; l2_tl(gatep, st, TD)[sc] = 0; }
            if (id < 0) {
                // an empty slot: a zero column that never moves (feasible set of slack 0, restart marked converged)
                l2_tl(slk, st, TD)[sc] = 0.0;
                L2_STORE_SETS(st, sc, 0.0)
            }
            if (id >= 0) atomicAdd(&ctl[0], 1);
            *(volatile rq_lds_int *)(l2_tl((int *)sy, st, TD) + sc) = (sc >= RQ_PARTS + NMW) ? 0x7fffffff : 0;
        }
        __syncthreads();
        if (ctl[0] == 0) break;                    // nothing left anywhere: done
        {
            // ---- build the columns of the restarts just taken: suggest(RANDOM) (qcqp.py:381-382: keyed normals, the stream
            // of randn_tiles_kernel) or the resident point, phase 1 (qcqp.py:101-149 through the visit of cd_phase1_sep.h), the
            // max violation = slack of phase 2 (qcqp.py:157) and the gate (qcqp.py:189).  The tiles live in global memory;
            // every thread only ever touches its own elements between two barriers.  Columns are named 16 tile + slot.
            // (priority over the neighbour workgroup's product streams: a double-precision instruction of the build otherwise
            //  waits for a whole matrix instruction of the other stream every time -- the build is latency, the streams have slack)
            __builtin_amdgcn_s_setprio(3);
            LG const CdLife *lf = l2_g(lifep);
            const long long pt0 = lf->prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
            const int lf_generate = lf->generate, lf_phase1 = lf->phase1;
            const double lf_viol_tol = lf->viol_tol;
            const int e0 = P.cptr[P.krep[0]];
            const double cp = P.cp[e0], cq = P.cq[e0], cr = P.cr[e0];
            const int rel = P.crel[e0];
            for (int c = 0; c < NS; c++) {
                const int ct = c >> 4, cc = c & 15;
                LG double *Xg = Xg0 + (int64_t)ct * n16 * 16;
                if (!l2_tl(snew, ct, TD)[cc]) {
                    if (l2_tl(sid, ct, TD)[cc] < 0) for (int64_t j = tid; j < n16; j += NT) Xg[j * 16 + cc] = 0.0;
                    continue;
                }
                if (!lf_generate) {
                    const int id = l2_tl(sid, ct, TD)[cc];
                    LG const double *src = l2_g(a0.b.X) + ((int64_t)(id >> 4) * n16) * 16 + (id & 15);
                    for (int64_t j0 = tid; j0 < n16; j0 += 4 * NT) {
                        double t4[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) t4[u] = (j0 + u * NT < n16) ? src[(j0 + u * NT) * 16] : 0.0;
#pragma unroll
                        for (int u = 0; u < 4; u++) if (j0 + u * NT < n16) Xg[(j0 + u * NT) * 16 + cc] = t4[u];
                    }
                }
            }
            if (lf_generate) {
                // keyed normals of the new columns, dealt to the waves in chunks of 64 element PAIRS (see phase 1 below: the waves
                // do not run at the same speed)
                if (tid == 0) {
                    int cnt = 0;
                    for (int k = 0; k < NS; k++) if (l2_tl(snew, k >> 4, TD)[k & 15]) p1cols[cnt++] = k;
                    ctl[4] = cnt; ctl[5] = 0;
                }
                __syncthreads();
                const int ncolg = ctl[4], nchg = (int)((n16 + 127) / 128);
                for (;;) {
                    int ch = 0;
                    if (lane == 0) ch = atomicAdd(&ctl[5], 1);
                    ch = __builtin_amdgcn_readfirstlane(ch);
                    if (ch >= ncolg * nchg) break;
                    const int c = p1cols[ch / nchg], ct = c >> 4, cc = c & 15;
                    LG double *Xg = Xg0 + (int64_t)ct * n16 * 16;
                    const int64_t j = (int64_t)(ch % nchg) * 128 + 2 * lane;
                    const uint64_t sd = l2_tl(sseed, ct, TD)[cc], gidx = l2_tl(sfirst, ct, TD)[cc] + (uint64_t)l2_tl(sid, ct, TD)[cc];
                    if (j < n16) {
                        double xo = 0.0;
                        const double xe = (j < P.n) ? l2_keyed_normal_pair(sd, gidx, (uint64_t)j, &xo) : 0.0;
                        Xg[j * 16 + cc] = xe;
                        Xg[(j + 1) * 16 + cc] = (j + 1 < P.n) ? xo : 0.0;
                    }
                }
            }
            __syncthreads();
            if (lf->prof && tid == 0) atomicAdd((unsigned long long *)lf->prof + 4, (unsigned long long)((long long)__builtin_amdgcn_s_memtime() - pt0));
            if (lf_phase1) {
                // The visits of a sweep are dealt to the waves in CHUNKS of 64 coordinates of one column from a counter in LDS:
                // the waves do not work at the same speed -- a double-precision instruction on SIMDs 1-3 waits for a slot between
                // the matrix instructions of the neighbouring workgroup's product streams, the wave on SIMD 0 (two chains, little
                // matrix work) runs several times faster -- and an even split made the build as slow as its slowest wave.  A visit's
                // result depends on (restart, coordinate, sweep) only: who computes it is immaterial.
                // (the Boolean family's class runs TWO visits per lane at once -- p1_band_visit_n: their dependency chains interleave)
                const bool band2 = !MULTI && cq == 0.0 && rel == RELOP_EQ && cp > 1e-4 && cr < -1e-3;      // workgroup-uniform
                const int cw = band2 ? 128 : 64;
                const int nch = (int)((P.n + cw - 1) / cw);
                for (int64_t t = 0; t < a.num_iters; t++) {
                    if (tid == 0) {
                        int cnt = 0;
                        for (int k = 0; k < NS; k++) if (l2_tl(snew, k >> 4, TD)[k & 15] && !l2_tl(p1fin, k >> 4, TD)[k & 15]) p1cols[cnt++] = k;
                        ctl[4] = cnt; ctl[5] = 0;
                    }
                    if (tid < NS) { l2_tl(p1key, tid >> 4, TD)[tid & 15] = l2_key(-QM_INF); l2_tl(p1upd, tid >> 4, TD)[tid & 15] = 0; }
                    __syncthreads();
                    const int ncol1 = ctl[4];
                    if (ncol1 == 0) break;
                    for (;;) {
                        int ch = 0;
                        if (lane == 0) ch = atomicAdd(&ctl[5], 1);
                        ch = __builtin_amdgcn_readfirstlane(ch);
                        if (ch >= ncol1 * nch) break;
                        const int c = p1cols[ch / nch], ct = c >> 4, cc = c & 15;
                        LG double *Xg = Xg0 + (int64_t)ct * n16 * 16;
                        const uint64_t sd = l2_tl(sseed, ct, TD)[cc], gidx = l2_tl(sfirst, ct, TD)[cc] + (uint64_t)l2_tl(sid, ct, TD)[cc];
                        double va = -QM_INF;
                        int fl = 0;
                        if (band2) {
                            const int64_t i2[2] = {(int64_t)(ch % nch) * 128 + lane, (int64_t)(ch % nch) * 128 + 64 + lane};
                            const bool on2[2] = {i2[0] < P.n, i2[1] < P.n};
                            double x2[2] = {on2[0] ? Xg[i2[0] * 16 + cc] : 1.0, on2[1] ? Xg[i2[1] * 16 + cc] : 1.0};
                            P1Visit V2[2];
                            p1_band_visit_n<2>(cp, cq, cr, i2, x2, on2, a.tol, lf_viol_tol, sd, gidx, t, V2);
#pragma unroll
                            for (int k = 0; k < 2; k++)
                                if (on2[k]) {
                                    if (V2[k].moved) { Xg[i2[k] * 16 + cc] = x2[k]; fl |= 1; }
                                    if (V2[k].status) fl |= (-V2[k].status) << 8;
                                    va = V2[k].vafter > va ? V2[k].vafter : va;
                                }
                        } else {
                            const int64_t i = (int64_t)(ch % nch) * 64 + lane;
                            if (i < P.n) {
                                if (MULTI) {          // the coordinate's own list of constraints (cd_phase1_sep.h, the serial path's visit)
                                    double xi = Xg[i * 16 + cc];
                                    P1Visit V;
                                    p1_sep_visit<MAXC>(P, i, xi, a.tol, lf_viol_tol, sd, gidx, t, V);
                                    fl = (V.moved ? 1 : 0) | ((-V.status) << 8);
                                    va = V.vafter;
                                    if (V.moved) Xg[i * 16 + cc] = xi;
                                } else {
                                    const double xi = l2_p1_visit(cp, cq, cr, rel, i, Xg[i * 16 + cc], a.tol, lf_viol_tol, sd, gidx, t, &fl, &va);
                                    if (fl & 1) Xg[i * 16 + cc] = xi;
                                }
                            }
                        }
                        const double vmax = l2_wave_max(va);
                        const bool anyupd = __builtin_amdgcn_ballot_w64((fl & 1) != 0) != 0ull;
                        const int st = (fl >> 8) ? -(fl >> 8) : 0;
                        if (lane == 0) { atomicMax(&l2_tl(p1key, ct, TD)[cc], l2_key(vmax)); if (anyupd) l2_tl(p1upd, ct, TD)[cc] = 1; }
                        if (st) l2_tl(p1st, ct, TD)[cc] = st;
                    }
                    __syncthreads();
                    if (tid < NS && l2_tl(snew, tid >> 4, TD)[tid & 15] && !l2_tl(p1fin, tid >> 4, TD)[tid & 15]) {
                        const int st = tid >> 4, sc = tid & 15;
                        l2_tl(p1sw, st, TD)[sc]++;
                        // done when feasible enough (qcqp.py:111); a sweep without any update is a fixed point of the map
                        if (l2_unkey(l2_tl(p1key, st, TD)[sc]) < lf_viol_tol || !l2_tl(p1upd, st, TD)[sc]) l2_tl(p1fin, st, TD)[sc] = 1;
                    }
                    __syncthreads();
                }
            }
            if (lf->prof && tid == 0) atomicAdd((unsigned long long *)lf->prof + 17, (unsigned long long)((long long)__builtin_amdgcn_s_memtime() - pt0));
            if (tid < NS) l2_tl(p1key, tid >> 4, TD)[tid & 15] = l2_key(-QM_INF);
            __syncthreads();
            for (int c = 0; c < NS; c++) {
                const int ct = c >> 4, cc = c & 15;
                if (!l2_tl(snew, ct, TD)[cc]) continue;
                LG const double *Xg = Xg0 + (int64_t)ct * n16 * 16;
                double v = -QM_INF;
                for (int64_t i = tid; i < P.n; i += NT) {
                    const double x = Xg[i * 16 + cc];
                    if (MULTI) {
                        for (int e = P.cptr[i]; e < P.cptr[i + 1]; e++) {
                            const double f = (P.cp[e] * x + P.cq[e]) * x + P.cr[e];
                            const double w = viol_of(f, P.crel[e]);
                            v = w > v ? w : v;
                        }
                    } else {
                        const double f = (cp * x + cq) * x + cr;
                        const double w = (rel == RELOP_EQ) ? fabs(f) : (f > 0.0 ? f : 0.0);
                        v = w > v ? w : v;
                    }
                }
                v = l2_wave_max(v);
                if (lane == 0) atomicMax(&l2_tl(p1key, ct, TD)[cc], l2_key(v));
            }
            __syncthreads();
            if (tid < NS && l2_tl(snew, tid >> 4, TD)[tid & 15]) {
                const int st = tid >> 4, sc = tid & 15;
                const double mvx = l2_unkey(l2_tl(p1key, st, TD)[sc]);
                l2_tl(slk, st, TD)[sc] = mvx;
                l2_tl(gatep, st, TD)[sc] = (mvx < lf_viol_tol && l2_tl(p1st, st, TD)[sc] == 0) ? 1 : 0;
                L2_STORE_SETS(st, sc, mvx)
            }
            __syncthreads();
            if (nlast < 16) {
                // n is not a multiple of 16: the padded coordinates of the last block (zero rows and columns of P0) hold a value
                // the step maps onto ITSELF -- the band's inner end / an end of the first interval / the highest end point --
                // so that they never move without a test in the chain's steps; zero again when the column is written out
                for (int w = tid; w < NS * (16 - nlast); w += NT) {
                    const int c = w % NS, ct = c >> 4, cc = c & 15, row = w / NS;
                    if (l2_tl(snew, ct, TD)[cc] || l2_tl(sid, ct, TD)[cc] < 0) {
                        const int nn = l2_tl(TC.n, ct, TD)[cc];
                        double xp = 0.0;
                        // (multi-class kinds: the padded coordinates count as class 0 -- the chain stages class 0 for them)
                        if (BK == L2_KIND_BAND) xp = nn >= 2 ? l2_tl(TC.lo, ct, TD)[16 * KCLV + cc] : 0.0;
                        else if (BK == L2_KIND_GEN) xp = nn >= 1 ? l2_tl(TC.hi, ct, TD)[cc] : 0.0;
                        else xp = nn >= 2 ? l2_tl(TC.hi, ct, TD)[16 * KCLV + cc] : l2_tl(TC.hi, ct, TD)[cc];
                        if (!(xp == xp) || __builtin_isinf(xp)) xp = 0.0;
                        (Xg0 + (int64_t)ct * n16 * 16)[(P.n + row) * 16 + cc] = xp;
                    }
                }
                __syncthreads();
            }
            if (LR) {
                // ---- factored objective: a column that starts here (or an empty slot) has Y = 0 and no pending moves; the chain's
                // loading sweep makes Y = L^T x0 of it
                const int r16 = 16 * a0.RB;
                for (int w = tid; w < NS * (32 + r16); w += NT) {
                    const int c = w % NS, ct = c >> 4, cc = c & 15, rowz = w / NS;
                    if (l2_tl(snew, ct, TD)[cc] || l2_tl(sid, ct, TD)[cc] < 0) {
                        if (rowz < 32) l2_tl(pend, ct, TD)[rowz * 16 + cc] = 0.0;
                        else l2_tl(ytile, ct, TD)[(rowz - 32) * 16 + cc] = 0.0;
                    }
                }
                __syncthreads();
            }
            if (lf->prof && tid == 0) {
                int nn = 0;
                for (int k = 0; k < NS; k++) nn += l2_tl(snew, k >> 4, TD)[k & 15] ? 1 : 0;
                atomicAdd((unsigned long long *)lf->prof + 0, (unsigned long long)((long long)__builtin_amdgcn_s_memtime() - pt0));
                atomicAdd((unsigned long long *)lf->prof + 2, 1ull);
                atomicAdd((unsigned long long *)lf->prof + 3, (unsigned long long)nn);
            }
            __builtin_amdgcn_s_setprio(0);
        }
        if (role == 0) {
            long long *cs_ = l2_tl(cst, rtile, TD);
            if (l2_tl(snew, rtile, TD)[r]) {
#pragma unroll
                for (int f = 0; f < 6; f++) cs_[f * 64 + lane] = 0;
                // the restart is not sweeping yet: gate not passed -> one frozen sweep (flag 1) that evaluates its objective;
                // passed, linear kind -> a frozen sweep first (flag 4) that evaluates f0 at the start of phase 2
                const int pass = l2_tl(gatep, rtile, TD)[r];
                cs_[6 * 64 + lane] = 0;
                cs_[4 * 64 + lane] = (pass && !PRE && !LR) ? 0 : 1;
                cs_[7 * 64 + lane] = (pass ? (PRE ? 4 : 0) : 1) | (LR ? 8 : 0);      // (factored objective: the loading sweep first)
                cs_[8 * 64 + lane] = 0;
            } else if (l2_tl(sid, rtile, TD)[r] < 0) {
                cs_[4 * 64 + lane] = 1; cs_[6 * 64 + lane] = 0;
                cs_[7 * 64 + lane] = 2; cs_[8 * 64 + lane] = 0;
            }
        }
        __syncthreads();
        if (tid == 0 && l2_g(lifep)->prof) *(long long *)(ctl + 6) = (long long)__builtin_amdgcn_s_memtime();

        // ================================================================ episode: the roles
        if (role > 0) {
            if (LR) l2_mfma_lr_role<NMW>(role - 1, rtile);
            else l2_mfma_role<NMW, CS>(role - 1, rtile);
        } else l2_chain_role<NMW, CS, KIND, TILES, LR>(rtile);

        __syncthreads();
        if (sy[L2_ABORT] || (TILES > 1 && l2_tl((int *)sy, 1, TD)[L2_ABORT])) {       // a wait gave up: unwind (the host reports it)
            if (tid == 0) __hip_atomic_store(l2_g(a0.abort), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        if (tid == 0 && l2_g(lifep)->prof)
        {
            const long long now_ = (long long)__builtin_amdgcn_s_memtime();
            atomicAdd((unsigned long long *)l2_g(lifep)->prof + 5, (unsigned long long)(now_ - *(long long *)(ctl + 6)));
            atomicAdd((unsigned long long *)l2_g(lifep)->prof + 20, (unsigned long long)(now_ - *(long long *)(ctl + 8)));    // from the chain's last interval to the barrier
        }
        // ================================================================ write out the slots that finished
        {
            // max violation of the final points, same expression as eval_kernel: (p x + q) x + r of the one constraint
            // every coordinate carries (single class, one constraint per coordinate)
            const int e0 = P.cptr[P.krep[0]];
            const double cp = P.cp[e0], cq = P.cq[e0], cr = P.cr[e0];
            const int rel = P.crel[e0];
            // the finished columns (two per tile and episode on average) are spread over ALL threads: item w = (column, row) with eight
            // loads in flight per thread (the tile lives in L2: a dependent load-store pair per row costs a round trip each, and with
            // one thread column per slot 14 of 16 threads had nothing to do); the max violation per column through LDS keys
            if (tid == 0) { int cnt = 0; for (int k = 0; k < NS; k++) if (l2_tl(sfin, k >> 4, TD)[k & 15]) p1cols[cnt++] = k; ctl[4] = cnt; }
            if (tid < NS) l2_tl(p1key, tid >> 4, TD)[tid & 15] = l2_key(-QM_INF);
            __syncthreads();
            const int nfin = ctl[4];
            const int64_t items = (int64_t)nfin * n16;
            for (int64_t w0 = tid; w0 < items; w0 += (int64_t)NT * 8) {
                double xv8[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int64_t w = w0 + (int64_t)NT * u;
                    const int col = p1cols[w < items ? w / n16 : 0];
                    const int64_t i = w % n16;
                    xv8[u] = (w < items && i < P.n) ? (Xg0 + (int64_t)(col >> 4) * n16 * 16)[i * 16 + (col & 15)] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int64_t w = w0 + (int64_t)NT * u;
                    if (w < items) {
                        const int col = p1cols[w / n16], id = l2_tl(sid, col >> 4, TD)[col & 15];
                        const int64_t i = w % n16;
                        l2_g(a0.b.X)[((int64_t)(id >> 4) * n16 + i) * 16 + (id & 15)] = xv8[u];
                        if (i < P.n) {
                            if (MULTI) {
                                double vv = -QM_INF;
                                for (int e = P.cptr[i]; e < P.cptr[i + 1]; e++) {
                                    const double w_ = viol_of((P.cp[e] * xv8[u] + P.cq[e]) * xv8[u] + P.cr[e], P.crel[e]);
                                    vv = w_ > vv ? w_ : vv;
                                }
                                atomicMax(&l2_tl(p1key, col >> 4, TD)[col & 15], l2_key(vv));
                            } else {
                                const double f = (cp * xv8[u] + cq) * xv8[u] + cr;
                                const double vv = (rel == RELOP_EQ) ? fabs(f) : (f > 0.0 ? f : 0.0);
                                atomicMax(&l2_tl(p1key, col >> 4, TD)[col & 15], l2_key(vv));
                            }
                        }
                    }
                }
            }
            __syncthreads();
            if (tid < NS && l2_tl(sfin, tid >> 4, TD)[tid & 15]) {
                const int st = tid >> 4, sc = tid & 15;
                const double mx = l2_unkey(l2_tl(p1key, st, TD)[sc]);
                const int id = l2_tl(sid, st, TD)[sc];
                l2_g(a0.b.visits)[id] = l2_tl(ovis, st, TD)[sc]; l2_g(a0.b.accepted)[id] = l2_tl(oacc, st, TD)[sc]; l2_g(a0.b.sweeps)[id] = l2_tl(oswp, st, TD)[sc];
                l2_g(a0.b.status)[id] = l2_tl(ost, st, TD)[sc];
                if (a0.b.f0out) l2_g(a0.b.f0out)[id] = l2_tl(of0, st, TD)[sc];
                if (a0.b.mvout) l2_g(a0.b.mvout)[id] = mx;
                LG const CdLife *lf = l2_g(lifep);
                l2_g(lf->sweeps1)[id] = l2_tl(p1sw, st, TD)[sc]; l2_g(lf->status1)[id] = l2_tl(p1st, st, TD)[sc];
                l2_g(lf->ran2)[id] = (uint8_t)l2_tl(gatep, st, TD)[sc];
                l2_tl(sid, st, TD)[sc] = -1; l2_tl(sfin, st, TD)[sc] = 0;
            }
            __syncthreads();
        }
    }
    if (tid0 == 0 && lf0->prof) {
        const unsigned long long dtl = (unsigned long long)((long long)__builtin_amdgcn_s_memtime() - life_t0);
        atomicAdd((unsigned long long *)lf0->prof + 1, dtl);
        atomicMax((unsigned long long *)lf0->prof + 16, dtl);
    }
}

#undef L2_STORE_SETS

// strictly upper triangle of the diagonal blocks (zeros elsewhere) and the per-block scalars the chain stages
__global__ void l2_pack_kernel(DevProblem P, double *Dpack, double *Spack) {
    const int b = blockIdx.x, t = threadIdx.x;            // 256 threads: entry (row t >> 4, column t & 15) of block b
    const int64_t n16 = P.n16;
    const int rr = t >> 4, cc = t & 15;
    Dpack[(int64_t)b * 256 + t] = (cc > rr) ? P.P0[(16 * (int64_t)b + rr) * n16 + 16 * b + cc] : 0.0;
    if (t < 16) {
        const int64_t i = 16 * (int64_t)b + t;
        const double d = P.P0[i * n16 + i], rc = P.rcp2d[i];
        Spack[(int64_t)b * 48 + t] = 0.5 * P.q0[i];
        Spack[(int64_t)b * 48 + 16 + t] = rc + rc;
        Spack[(int64_t)b * 48 + 32 + t] = d;
    }
}

// fragments of the factor L (n16 x r16, row-major, zero-padded) for the products (Gpack) and the updates of Y (Upack): see l2_mfma_lr_role
__global__ void l2_pack_factor_kernel(const double *__restrict__ L, double *__restrict__ Gpack, double *__restrict__ Upack, int NB, int RB) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)NB * RB * 256) return;
    const int e = (int)(idx & 1), lane = (int)((idx >> 1) & 63), vp = (int)((idx >> 7) & 1);
    const int64_t rest = idx >> 8;
    const int beta = (int)(rest % RB), b = (int)(rest / RB), v = 2 * vp + e;
    const int64_t r16 = 16 * (int64_t)RB;
    Gpack[idx] = L[(16 * (int64_t)b + (lane & 15)) * r16 + 16 * beta + 4 * v + (lane >> 4)];
    Upack[idx] = L[(16 * (int64_t)b + 4 * v + (lane >> 4)) * r16 + 16 * beta + (lane & 15)];
}

template <int NMW, int CS, int TILES, int LR>
int l2_launch_kind(const CdLife2Args &a, int kind, int wgs, size_t lds, hipStream_t st) {
    constexpr bool MK = TILES == 1 && !LR;        // the multi-class kinds exist for one tile per workgroup, without an objective factor
    if (!MK && (kind == L2_KIND_GENK || kind == L2_KIND_LINK)) return (int)hipErrorInvalidValue;
    auto k = kind == L2_KIND_BAND ? cd_life_kernel<NMW, CS, L2_KIND_BAND, TILES, LR> : (kind == L2_KIND_GEN || LR) ? cd_life_kernel<NMW, CS, L2_KIND_GEN, TILES, LR>
             : kind == L2_KIND_GENK ? cd_life_kernel<NMW, CS, MK ? L2_KIND_GENK : L2_KIND_GEN, TILES, LR>
             : kind == L2_KIND_LINK ? cd_life_kernel<NMW, CS, MK ? L2_KIND_LINK : L2_KIND_GEN, TILES, LR>
                                    : cd_life_kernel<NMW, CS, LR ? L2_KIND_GEN : L2_KIND_LIN, TILES, LR>;
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k, dim3((unsigned)wgs), dim3((NMW == 3 && TILES == 1) ? 256 : 512), lds, st, a);
    return (int)hipGetLastError();
}

}  // namespace

size_t cd_life2_lds_bytes(int nmw, int cs, int tiles, int lr, int kind) {
    const bool multi = kind == L2_KIND_GENK || kind == L2_KIND_LINK;
    return ((size_t)L2_SHARED_DOUBLES + (size_t)tiles * (size_t)l2_tile_doubles(nmw, cs > 0 ? cs : 1, lr, multi ? L2_KCL : 1, multi ? 2 : 1)) * sizeof(double) + 256;
}

// can the factored-objective kernel take a factor of r columns for this problem?  (rank, the scratch of the column build)
bool cd_life2_factor_ok(const DevProblem &P, int64_t r) {
    const int64_t r16 = (r + 15) / 16 * 16;
    if (r < 1 || r16 > 16 * L2_YBMAX || P.NB < 8) return false;
    return true;
}

int cd_life2_pack_factor(const double *Lrow, double *Gpack, double *Upack, int NB, int RB, hipStream_t st) {
    const int64_t tot = (int64_t)NB * RB * 256;
    hipLaunchKernelGGL(l2_pack_factor_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, Lrow, Gpack, Upack, NB, RB);
    return (int)hipGetLastError();
}

int cd_life2_max_wgs(int nmw, int cus, int tiles) { return (nmw == 3 && tiles == 1) ? 2 * cus : cus; }

bool cd_life2_config(const DevProblem &P, int Kreal, int objclass, bool symcls, int factor_rb, int *nmw, int *cs, int *kind) {
    // one class with one constraint per coordinate: the BAND / GEN / LIN kinds; up to L2_KCL classes with up to two constraints per
    // coordinate (every real coordinate constrained): GENK / LINK
    if (!P.sep || P.maxc < 1 || P.maxc > 2 || Kreal < 1 || Kreal > L2_KCL) return false;
    const bool multi = P.maxc > 1 || Kreal > 1;
    if (objclass != 1 && objclass != 2) return false;
    const int NB = (int)P.NB;
    if (NB < 3) return false;
    int w, c;
    if (NB <= 4) { w = 3; c = 0; }
    else if (NB - 4 <= 3 * RQ_MAXU) { w = 3; c = (NB < 8) ? 2 : 4; }
    else if (factor_rb > 0 && !multi && objclass == 1 && NB <= 256) { w = 3; c = 0; }      // factored: nothing in its registers grows with n
    else if (NB - 4 <= 7 * RQ_MAXU) { w = 7; c = 4; }
    else return false;
    *nmw = w; *cs = c;
    *kind = multi ? (objclass == 2 ? L2_KIND_LINK : L2_KIND_GENK) : objclass == 2 ? L2_KIND_LIN : (symcls ? L2_KIND_BAND : L2_KIND_GEN);
    return true;
}

// tiles per workgroup (0 = automatic; QCQPMI_L2_TILES overrides: experiments).  Two tiles per workgroup -- one eight-wave workgroup
// per CU, joint episodes, the build by 512 threads -- exist for the factored objective only and are NOT the default: measured at
// n = 1024 (20 x 4096 restarts) 33.7-35.4 ms against 32.6-34.4 ms for two four-wave workgroups per CU, whose column builds overlap
// the neighbour's products (profiles/r06_summary.md; without a factor: 52.0-53.5 against 49.5-51.0 ms,
// profiles/r06_headline_experiments.md -- that instantiation was removed).
int cd_life2_tiles(const DevProblem &P, int nmw, int cs, int64_t restarts, int cus, int requested, int lr) {
    (void)P; (void)restarts; (void)cus;
    if (!lr || nmw != 3 || cs != 0) return 1;
    int t = requested;
    if (const char *ev = getenv("QCQPMI_L2_TILES")) t = atoi(ev);
    return t == 2 ? 2 : 1;
}

int cd_life2_pack(const DevProblem &P, double *Dpack, double *Spack, hipStream_t st) {
    hipLaunchKernelGGL(l2_pack_kernel, dim3((unsigned)P.NB), dim3(256), 0, st, P, Dpack, Spack);
    return (int)hipGetLastError();
}

int cd_life2_launch(const CdLife2Args &a, int nmw, int cs, int kind, int tiles, int wgs, hipStream_t st) {
    const int lr = a.RB > 0 ? 1 : 0;
    if (const char *ev = getenv("QCQPMI_L2_CS")) {       // experiments: the chain's share (blocks of the contraction), four-wave workgroups only
        const int k2 = atoi(ev);
        if (!lr && nmw == 3 && tiles == 1 && (k2 == 0 || k2 == 2 || k2 == 4) && k2 + 3 <= (int)a.P.NB && (int)a.P.NB - k2 <= 3 * RQ_MAXU) cs = k2;
    }
    const size_t lds = cd_life2_lds_bytes(nmw, cs, tiles, lr, kind);
    if (getenv("QCQPMI_L2_DEBUG")) {
        int occ = -1;
        hipError_t e = lr ? (tiles == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, cd_life_kernel<3, 0, L2_KIND_BAND, 2, 1>, 512, lds)
                                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, cd_life_kernel<3, 0, L2_KIND_BAND, 1, 1>, 256, lds))
                     : nmw == 3 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, cd_life_kernel<3, 4, L2_KIND_BAND, 1, 0>, 256, lds)
                                : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, cd_life_kernel<7, 4, L2_KIND_BAND, 1, 0>, 512, lds);
        fprintf(stderr, "cd_life2_launch: nmw %d cs %d kind %d tiles %d lr %d wgs %d lds %zu: occupancy %d workgroups per CU (%s)\n", nmw, cs, kind, tiles, lr, wgs, lds, occ, hipGetErrorString(e));
    }
    if (wgs < 1) wgs = 1;
    if (lr) {
        if (nmw != 3 || cs != 0 || kind == L2_KIND_LIN) return (int)hipErrorInvalidValue;
        return tiles == 2 ? l2_launch_kind<3, 0, 2, 1>(a, kind, wgs, lds, st) : l2_launch_kind<3, 0, 1, 1>(a, kind, wgs, lds, st);
    }
    if (tiles != 1) return (int)hipErrorInvalidValue;
    if (nmw == 3) {
        if (cs == 0) return l2_launch_kind<3, 0, 1, 0>(a, kind, wgs, lds, st);
        if (cs == 2) return l2_launch_kind<3, 2, 1, 0>(a, kind, wgs, lds, st);
        if (cs == 4) return l2_launch_kind<3, 4, 1, 0>(a, kind, wgs, lds, st);
        return (int)hipErrorInvalidValue;
    }
    if (nmw == 7 && cs == 4) return l2_launch_kind<7, 4, 1, 0>(a, kind, wgs, lds, st);
    return (int)hipErrorInvalidValue;
}

const char *cd_life2_name(int nmw, int kind, int tiles, int lr) {
    if (kind == L2_KIND_GENK) return nmw == 3 ? "cd_life_kernel<3,gen,classes>" : "cd_life_kernel<7,gen,classes>";
    if (kind == L2_KIND_LINK) return nmw == 3 ? "cd_life_kernel<3,lin,classes>" : "cd_life_kernel<7,lin,classes>";
    if (lr) return tiles == 2 ? (kind == L2_KIND_BAND ? "cd_life_kernel<3,band,2 tiles,factored>" : "cd_life_kernel<3,gen,2 tiles,factored>")
                              : (kind == L2_KIND_BAND ? "cd_life_kernel<3,band,factored>" : "cd_life_kernel<3,gen,factored>");
    if (tiles == 2) return kind == L2_KIND_BAND ? "cd_life_kernel<3,band,2 tiles>" : kind == L2_KIND_GEN ? "cd_life_kernel<3,gen,2 tiles>" : "cd_life_kernel<3,lin,2 tiles>";
    if (nmw == 3) return kind == L2_KIND_BAND ? "cd_life_kernel<3,band>" : kind == L2_KIND_GEN ? "cd_life_kernel<3,gen>" : "cd_life_kernel<3,lin>";
    return kind == L2_KIND_BAND ? "cd_life_kernel<7,band>" : kind == L2_KIND_GEN ? "cd_life_kernel<7,gen>" : "cd_life_kernel<7,lin>";
}

}  // namespace qcqpmi
