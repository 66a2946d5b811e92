// cd_phase2_qs_kernel -- coordinate descent phase 2 (qcqp.py:152-178) for the Boolean family with RESTART-LEVEL scheduling:
// the pipelined kernel of cd_phase2_q.h (same roles, same sync protocol, same per-coordinate arithmetic: see there and
// DESIGN.md section 4.1) wrapped into EPISODES.  A workgroup owns 16 slots of its X tile in LDS; an episode runs the role
// code from a sweep boundary until, at a later sweep boundary, some slot's restart is done (converged, qcqp.py:172-176, or at
// the sweep limit) and the device-side queue still has restarts -- then the finished columns are written out (point,
// tracked objective, max violation, counters), the free slots take the next restarts of the queue, and the next episode
// starts; restarts that are not done simply continue (their state lives in the chain wave's registers and in the tile).
// Lifecycle mode (round 4, cd_queue.h): the queue runs over the restarts of SEVERAL populations and a slot builds its restart
// itself -- normals, phase 1, gate -- before phase 2.  (Rounds 3's chained launches and ring mode -- one persistent launch on
// a CU-masked stream serving the populations of several contexts -- were removed in round 4: the lifecycle mode does what
// they were after without a CU partition, extra hardware queues or a second stream; DESIGN.md section 4.1c keeps the record.)
//
// Determinism.  A restart's values must not depend on where the episode boundaries fall (they depend on the other slots).
// Every product of the kernel is therefore summed in ONE association: the multiplying waves always leave out the two
// blocks a sweep rewrote last and the chain wave always supplies them -- at the start of an episode from a virtual
// interval that recomputes, from the tile, exactly what the end of a sweep leaves in its registers.  (cd_phase2_q_kernel
// sums the first two products of a launch differently: the two kernels agree to rounding, not bit for bit.)
#include "cd_queue.h"

#include "onevar.h"
#include "cd_phase1_sep.h"

namespace qcqpmi {
// (cd_phase2.h, which the role helpers come with, expects the MFMA building block of kernels.hip to be declared)
typedef double v4d __attribute__((ext_vector_type(4)));
template <typename XPtr>
__device__ inline v4d block_rows_times_X(const double *__restrict__ Ab, XPtr Xs, int kk0, int kk1, int lane, v4d acc) {
    const int xoff = (lane >> 4) * 16 + (lane & 15);
    for (int kk = kk0; kk < kk1; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ab[(int64_t)kk * 64 + lane], Xs[kk * 64 + xoff], acc, 0, 0, 0);
    return acc;
}
}  // namespace qcqpmi

#include "cd_phase2_q.h"

namespace qcqpmi {
namespace {

__device__ inline int qs_load_int(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double qs_load_d(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// pointers typed as global memory: a generic pointer makes the compiler emit FLAT loads / stores / atomics, which count against
// the LDS counter as well -- and the roles of this kernel synchronise through LDS words (measured in round 3: 30 % slower)
#define QG __attribute__((address_space(1)))
struct CdBatchG {
    QG double *X;
    QG const double *f0cur, *slack;
    QG const uint8_t *flag;
    QG int64_t *visits, *accepted, *sweeps;
    QG int *status;
    QG double *f0out, *mvout;
    int64_t R;
    uint64_t seed, first_index;
    QG int *next;
};
template <class T>
__device__ __attribute__((always_inline)) inline QG T *qs_g(T *p) { return (QG T *)p; }
__device__ __attribute__((always_inline)) inline CdBatchG qs_batch(const CdBatch &t) {
    CdBatchG b;
    b.X = qs_g(t.X); b.f0cur = qs_g(t.f0cur); b.slack = qs_g(t.slack); b.flag = qs_g(t.flag);
    b.visits = qs_g(t.visits); b.accepted = qs_g(t.accepted); b.sweeps = qs_g(t.sweeps); b.status = qs_g(t.status);
    b.f0out = qs_g(t.f0out); b.mvout = qs_g(t.mvout); b.next = qs_g(t.next);
    b.R = t.R; b.seed = t.seed; b.first_index = t.first_index;
    return b;
}
__device__ inline int qs_load_int(QG const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double qs_load_d(QG const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline int qs_add(QG int *p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// order-preserving map double -> u64 and back (LDS integer atomics as max-reductions over the threads)
__device__ inline unsigned long long qs_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double qs_unkey(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
__device__ inline double qs_wave_max(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const double w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}

// Lifecycle mode: the heavy scalar code of a restart's start -- Box-Muller normals, the phase-1 visit (bisection, Philox,
// square roots) -- as REAL FUNCTION CALLS.  Inlined into the kernel they cost it 150 VGPR spills in the multiplying waves'
// product loop (the kernel sits exactly at the 256-register limit there); called, they have register allocations of their own.
__device__ __attribute__((noinline)) double qs_keyed_normal(uint64_t seed, uint64_t restart, uint64_t elem) {
    return keyed_normal(seed, restart, elem);
}
// the normals of the element pair (elem, elem + 1), elem even: keyed_normal's own expressions (philox.h) -- the two share the
// counter block elem >> 1, the radius and the angle; the even element takes the cosine, the odd one the sine -- evaluated once
__device__ __attribute__((noinline)) double qs_keyed_normal_pair(uint64_t seed, uint64_t restart, uint64_t elem, double *odd) {
    const U4 o = philox4x32_10((uint32_t)(elem >> 1), (uint32_t)(elem >> 33), 0xA5A50000u, (uint32_t)restart, (uint32_t)seed,
                               (uint32_t)(seed >> 32) ^ (uint32_t)(restart >> 32));
    const double u1 = (((double)(o.x >> 5) * 67108864.0 + (double)(o.y >> 6)) + 0.5) / 9007199254740992.0;
    const double u2 = u53(o.z, o.w);
    const double rad = sqrt(-2.0 * log(u1));
    const double ang = 6.283185307179586476925286766559 * u2;
    *odd = rad * sin(ang);
    return rad * cos(ang);
}
// one phase-1 visit of coordinate i (value x) of a problem whose coordinates all carry the one constraint (p, q, r, relop);
// returns the new value, *flags: bit 0 moved, bits 8.. = -status; *vafter: the constraint's violation afterwards
__device__ __attribute__((noinline)) double qs_p1_visit(double p, double q, double r, int relop, int64_t i, double x, double tol,
                                                        double viol_tol, uint64_t seed, uint64_t restart, int64_t t, int *flags,
                                                        double *vafter) {
    P1Visit V;
    if (q == 0.0 && relop == RELOP_EQ && p > 1e-4) {
        p1_band_visit(p, q, r, i, x, tol, viol_tol, seed, restart, t, V);      // the class of the headline family, resolved by hand
    } else {
        const double cp[1] = {p}, cq[1] = {q}, cr[1] = {r};
        const int crel[1] = {relop};
        p1_sep_visit_core<1>(1, cp, cq, cr, crel, i, x, tol, viol_tol, seed, restart, t, V);
    }
    *flags = (V.moved ? 1 : 0) | ((-V.status) << 8);
    *vafter = V.vafter;
    return x;
}

// two visits of the Boolean family's class at once (round 5, cd_phase1_sep.h::p1_band_visit_n: the dependency chains of two visits
// interleave; element k of the result is the single visit's, bit for bit) -- a call like the one above, for the same reason
__device__ __attribute__((noinline)) void qs_p1_visit2(double p, double q, double r, int64_t i0, int64_t i1, bool on1, double *x0, double *x1,
                                                       double tol, double viol_tol, uint64_t seed, uint64_t restart, int64_t t, int *flags,
                                                       double *vafter) {
    const int64_t ii[2] = {i0, i1};
    const bool on[2] = {true, on1};
    double xx[2] = {*x0, *x1};
    P1Visit V[2];
    p1_band_visit_n<2>(p, q, r, ii, xx, on, tol, viol_tol, seed, restart, t, V);
    *x0 = xx[0]; *x1 = xx[1];
    int fl = (V[0].moved ? 1 : 0) | ((on1 && V[1].moved) ? 2 : 0);
    if (V[0].status) fl |= (-V[0].status) << 8;
    else if (on1 && V[1].status) fl |= (-V[1].status) << 8;
    *flags = fl;
    *vafter = on1 ? (V[0].vafter > V[1].vafter ? V[0].vafter : V[1].vafter) : V[0].vafter;
}

// CS: blocks of the contraction the chain wave multiplies itself (0..RQ_CSMAX); LIFE: lifecycle mode (cd_queue.h)
template <int CS, bool LIFE>
__global__ __launch_bounds__(512) void cd_phase2_qs_kernel(CdQueueArgs a0) {
    const CdQueueArgs &a = a0;
    constexpr int MAXC = 1;
    constexpr int CSU = CS > 0 ? CS : 1;
    extern __shared__ double smem[];
    const DevProblem &P = a.P;
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int64_t n16 = P.n16;
    const int NB = (int)P.NB, KS = (int)P.KS;
    // ---- dynamic LDS carve-up (cd_phase2_q_kernel's, plus the slot tables)
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
    rq_lds_int *sy = (rq_lds_int *)(int *)sp; sp += 8;     // synchronisation words (16 ints, 16-byte aligned)
    double *f0new = sp; sp += 16;                  // objective at phase-2 start of the restarts just taken
    double *of0 = sp; sp += 16;                    // outputs of the slots that finished in the episode
    long long *ovis = (long long *)sp; sp += 16;
    long long *oacc = (long long *)sp; sp += 16;
    long long *oswp = (long long *)sp; sp += 16;
    int *sid = (int *)sp; sp += 8;                 // restart held by a slot (-1: none)
    int *snew = (int *)sp; sp += 8;                // slot refilled before this episode
    int *sfin = (int *)sp; sp += 8;                // slot's restart finished in this episode
    int *ost = (int *)sp; sp += 8;
    int *ctl = (int *)sp; sp += 8;                 // [0] occupied slots
    long long *cst = (long long *)sp; sp += 64 * 10; // the chain wave's per-lane state between episodes: [field][lane]
    unsigned long long *sseed = (unsigned long long *)sp; sp += 16;     // seed / first global index (minus the slot's queue index) of the slot's population
    unsigned long long *sfirst = (unsigned long long *)sp; sp += 16;
    // lifecycle mode: phase 1 of the restarts just taken -- per slot: max-violation key (reduction), "a coordinate moved",
    // finished, sweeps, status; passed the gate
    unsigned long long *p1key = (unsigned long long *)sp; sp += 16;
    int *p1upd = (int *)sp; sp += 8;
    int *p1fin = (int *)sp; sp += 8;
    int *p1sw = (int *)sp; sp += 8;
    int *p1st = (int *)sp; sp += 8;
    int *gatep = (int *)sp; sp += 8;

    // ---- the chain wave's per-restart state (lane 4 r + g: restart slot r) is parked in LDS between episodes: fields
    // 0 upd_counter, 1 visits, 2 accepted, 3 sweeps, 4 conv, 5 status, 6 fpart (bits), 7 lifecycle flags (1 frozen sweep, 2 done),
    // 8 window sum (bits); inside an episode it lives in the
    // chain role's registers like in cd_phase2_q_kernel (keeping it in registers of the whole kernel made the multiplying
    // waves spill)
    if (tid0 < 64) {
#pragma unroll
        for (int f = 0; f < 10; f++) cst[f * 64 + tid0] = (f == 4) ? 1 : 0;
    }
    if (tid0 < 16) { sid[tid0] = -1; sfin[tid0] = 0; }
    const long long life_t0 = (LIFE && tid0 == 0) ? (long long)__builtin_amdgcn_s_memtime() : 0;
    __syncthreads();
    const int64_t gmax = (int64_t)1 << 40;         // the roles end through RQ_STOP

    for (;;) {
        // The lane index is made opaque per episode: otherwise the compiler hoists every lane-dependent address of every
        // role out of the episode loop and keeps them all alive through all roles (70+ VGPR spills in the chain's loop).
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const CdLife *lifep = a0.life;               // (opaque per episode as well: its fields are read where they are used)
        asm volatile("" : "+s"(lifep));
        const int lane = tid & 63, r = lane >> 2, gq = lane & 3;
        // ================================================================ refill: free slots take the next restarts
        if (tid == 0) ctl[0] = 0;
        __syncthreads();
        if (tid < 16) {
            int id = sid[tid], nw = 0;
            if (id < 0 && LIFE) {
                // lifecycle mode: the queue is a counter over all restarts of the run; the column is built below
                QG const CdLife *lf = qs_g(lifep);
                const int idx = qs_add(qs_g(a0.b.next), 1);
                if (idx < (int)lf->Rtotal) {
                    id = idx; nw = 1;
                    const uint64_t pop = (uint64_t)idx / (uint64_t)lf->Rpop, rho = (uint64_t)idx % (uint64_t)lf->Rpop;
                    sseed[tid] = lf->seed + pop * lf->seed_stride;
                    sfirst[tid] = lf->first_index + pop * lf->first_stride + rho - (uint64_t)idx;    // + id = the global restart index
                }
            } else if (id < 0) {
                const CdBatchG B = qs_batch(a0.b);
                for (;;) {
                    const int idx = qs_add(B.next, 1);      // (runs past R by at most 16 per workgroup and episode: harmless)
                    if (idx >= (int)B.R) break;
                    if (B.flag[idx]) {                      // passed the gate of improve_coord_descent (qcqp.py:189)
                        id = idx; nw = 1;
                        sseed[tid] = B.seed; sfirst[tid] = B.first_index;
                        break;
                    }
                }
            }
            sid[tid] = id; snew[tid] = nw;
            if (LIFE) { if (nw) { p1fin[tid] = 0; p1sw[tid] = 0; p1st[tid] = 0; gatep[tid] = 0; } }
            if (nw && !LIFE) {
                const CdBatchG B = qs_batch(a0.b);
                slk[tid] = B.slack[id];
                f0new[tid] = B.f0cur[id];
                FeasSet<MAXC> C;
                compute_set<MAXC>(P, P.krep[0], slk[tid], C);
                store_set<MAXC>(TC, tid, C);
            } else if (id < 0) {
                // an empty slot: a zero column that never moves (feasible set of slack 0, restart marked converged)
                slk[tid] = 0.0;
                FeasSet<MAXC> C;
                compute_set<MAXC>(P, P.krep[0], 0.0, C);
                store_set<MAXC>(TC, tid, C);
            }
            if (id >= 0) atomicAdd(&ctl[0], 1);
        }
        if (tid < 16) *(volatile rq_lds_int *)(sy + tid) = (tid >= RQ_PARTS && tid < RQ_PARTS + 3) ? -1 : 0;   // words of the even-product waves start odd
        __syncthreads();
        if (ctl[0] == 0) break;                    // nothing left anywhere: done
        if (LIFE) {
            // ---- lifecycle mode: build the columns of the restarts just taken -- suggest(RANDOM) (qcqp.py:381-382: keyed
            // normals, the stream of randn_tiles_kernel), phase 1 (qcqp.py:101-149 through p1_sep_visit: the moves of
            // cd_phase1_sep_kernel bit for bit), the max violation = slack of phase 2 (qcqp.py:157) and the gate (qcqp.py:189)
            QG const CdLife *lf = qs_g(lifep);
            const long long pt0 = lf->prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
            const int lf_generate = lf->generate, lf_phase1 = lf->phase1;
            const double lf_viol_tol = lf->viol_tol;
            const int e0 = P.cptr[P.krep[0]];
            const double cp = P.cp[e0], cq = P.cq[e0], cr = P.cr[e0];
            const int rel = P.crel[e0];
            for (int c = 0; c < 16; c++) {
                if (!snew[c]) {
                    if (sid[c] < 0) for (int64_t j = tid; j < n16; j += 512) Xs[j * 16 + c] = 0.0;
                    continue;
                }
                if (lf_generate) {
                    const uint64_t sd = sseed[c], gidx = sfirst[c] + (uint64_t)sid[c];
                    for (int64_t j = 2 * (int64_t)tid; j < n16; j += 1024) {      // n is a multiple of 16 here: pairs never straddle n
                        double xo = 0.0;
                        const double xe = (j < P.n) ? qs_keyed_normal_pair(sd, gidx, (uint64_t)j, &xo) : 0.0;
                        Xs[j * 16 + c] = xe;
                        Xs[(j + 1) * 16 + c] = xo;
                    }
                } else {
                    QG const double *src = qs_g(a0.b.X) + ((int64_t)(sid[c] >> 4) * n16) * 16 + (sid[c] & 15);
                    for (int64_t j = tid; j < n16; j += 512) Xs[j * 16 + c] = src[j * 16];
                }
            }
            __syncthreads();
            if (lf->prof && tid == 0) atomicAdd((unsigned long long *)lf->prof + 4, (unsigned long long)((long long)__builtin_amdgcn_s_memtime() - pt0));
            if (lf_phase1) {
                for (int64_t t = 0; t < a.num_iters; t++) {
                    if (tid == 0) { int cnt = 0; for (int k = 0; k < 16; k++) cnt += (snew[k] && !p1fin[k]) ? 1 : 0; ctl[4] = cnt; }
                    if (tid < 16) { p1key[tid] = qs_key(-QM_INF); p1upd[tid] = 0; }
                    __syncthreads();
                    if (ctl[4] == 0) break;
                    for (int c = 0; c < 16; c++) {
                        if (!snew[c] || p1fin[c]) continue;       // workgroup-uniform
                        const uint64_t sd = sseed[c], gidx = sfirst[c] + (uint64_t)sid[c];
                        double vmax = -QM_INF;
                        int upd = 0, st = 0;
                        if (cq == 0.0 && rel == RELOP_EQ && cp > 1e-4 && cr < -1e-3) {       // the Boolean family: two visits per call
                            for (int64_t i = tid; i < P.n; i += 1024) {
                                int fl;
                                double va;
                                const int64_t i1 = i + 512;
                                const bool on1 = i1 < P.n;
                                double xa = Xs[i * 16 + c], xb = on1 ? Xs[i1 * 16 + c] : 1.0;
                                qs_p1_visit2(cp, cq, cr, i, i1, on1, &xa, &xb, a.tol, lf_viol_tol, sd, gidx, t, &fl, &va);
                                if (fl >> 8) st = -(fl >> 8);
                                if (fl & 1) { Xs[i * 16 + c] = xa; upd = 1; }
                                if (fl & 2) { Xs[i1 * 16 + c] = xb; upd = 1; }
                                vmax = va > vmax ? va : vmax;
                            }
                        } else
                        for (int64_t i = tid; i < P.n; i += 512) {
                            int fl;
                            double va;
                            const double xi = qs_p1_visit(cp, cq, cr, rel, i, Xs[i * 16 + c], a.tol, lf_viol_tol, sd, gidx, t, &fl, &va);
                            if (fl >> 8) st = -(fl >> 8);
                            if (fl & 1) { Xs[i * 16 + c] = xi; upd = 1; }
                            vmax = va > vmax ? va : vmax;
                        }
                        vmax = qs_wave_max(vmax);
                        if (lane == 0) atomicMax(&p1key[c], qs_key(vmax));
                        if (upd) p1upd[c] = 1;
                        if (st) p1st[c] = st;
                    }
                    __syncthreads();
                    if (tid < 16 && snew[tid] && !p1fin[tid]) {
                        p1sw[tid]++;
                        // done when feasible enough (qcqp.py:111); a sweep without any update is a fixed point of the map
                        if (qs_unkey(p1key[tid]) < lf_viol_tol || !p1upd[tid]) p1fin[tid] = 1;
                    }
                    __syncthreads();
                }
            }
            if (tid < 16) p1key[tid] = qs_key(-QM_INF);
            __syncthreads();
            for (int c = 0; c < 16; c++) {
                if (!snew[c]) continue;
                double v = -QM_INF;
                for (int64_t i = tid; i < P.n; i += 512) {
                    const double x = Xs[i * 16 + c];
                    const double f = (cp * x + cq) * x + cr;
                    const double w = (rel == RELOP_EQ) ? fabs(f) : (f > 0.0 ? f : 0.0);
                    v = w > v ? w : v;
                }
                v = qs_wave_max(v);
                if (lane == 0) atomicMax(&p1key[c], qs_key(v));
            }
            __syncthreads();
            if (tid < 16 && snew[tid]) {
                const double mvx = qs_unkey(p1key[tid]);
                slk[tid] = mvx;
                gatep[tid] = (mvx < lf_viol_tol && p1st[tid] == 0) ? 1 : 0;
                FeasSet<MAXC> C;
                compute_set<MAXC>(P, P.krep[0], mvx, C);
                store_set<MAXC>(TC, tid, C);
            }
            __syncthreads();
            if (lf->prof && tid == 0) {
                int nn = 0;
                for (int k = 0; k < 16; k++) nn += snew[k] ? 1 : 0;
                atomicAdd((unsigned long long *)lf->prof + 0, (unsigned long long)((long long)__builtin_amdgcn_s_memtime() - pt0));
                atomicAdd((unsigned long long *)lf->prof + 2, 1ull);
                atomicAdd((unsigned long long *)lf->prof + 3, (unsigned long long)nn);
            }
        } else {
            // columns of the restarts just taken (sc1 loads: the next population was written by kernels of another stream
            // while this one was running), zero columns for empty slots
            const int col = tid & 15;
            const int id = sid[col];
            if (snew[col]) {
                // eight loads in flight per thread
                QG const double *src = qs_g(a0.b.X) + ((int64_t)(id >> 4) * n16) * 16 + (id & 15);
                for (int64_t j0 = tid >> 4; j0 < n16; j0 += 32 * 8) {
                    double pv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int64_t j = j0 + 32 * u;
                        pv[u] = (j < n16) ? src[j * 16] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) { const int64_t j = j0 + 32 * u; if (j < n16) Xs[j * 16 + col] = pv[u]; }
                }
            } else if (id < 0) {
                for (int64_t j = tid >> 4; j < n16; j += 32) Xs[j * 16 + col] = 0.0;
            }
        }
        if (wave == 0) {
            if (snew[r]) {
#pragma unroll
                for (int f = 0; f < 6; f++) cst[f * 64 + lane] = 0;
                if (LIFE) {
                    // the objective is tracked RELATIVE to the start of phase 2 (its value comes from the converged window or
                    // from a frozen sweep, see the chain role); a restart that did not pass the gate only takes a frozen sweep
                    cst[6 * 64 + lane] = 0;
                    cst[4 * 64 + lane] = gatep[r] ? 0 : 1;
                    cst[7 * 64 + lane] = gatep[r] ? 0 : 1;
                    cst[8 * 64 + lane] = 0;
                } else {
                    cst[6 * 64 + lane] = __double_as_longlong((gq == 0) ? f0new[r] : 0.0);
                }
            } else if (sid[r] < 0) {
                cst[4 * 64 + lane] = 1; cst[6 * 64 + lane] = 0;
                if (LIFE) { cst[7 * 64 + lane] = 2; cst[8 * 64 + lane] = 0; }
            }
        }
        __syncthreads();
        if (LIFE && tid == 0 && qs_g(lifep)->prof) *(long long *)(ctl + 6) = (long long)__builtin_amdgcn_s_memtime();     // (stashed in LDS: not a register of the chain wave)

        // ================================================================ episode: the roles of cd_phase2_q_kernel
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
                    d4[e] = P.P0[(16 * (int64_t)bn + (st >> 4)) * n16 + 16 * bn + (st & 15)];
                }
                if (lane < 16) { sq = P.q0[16 * (int64_t)bn + lane]; sr = P.rcp2d[16 * (int64_t)bn + lane];
                                 sd = P.P0[(16 * (int64_t)bn + lane) * n16 + 16 * bn + lane]; }
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
            // No address arithmetic and no LDS traffic in the product loop.  Unit u of this SIMD is block sm + 3 u:
            //   B operands: PERSISTENT in registers (4 per unit); a product only re-reads the (at most two) blocks committed
            //   since this wave's previous product;
            //   A fragments of block row `row`: buffer loads, descriptor = P.Apack2, scalar offset = row * KS * 512 + block * 2048
            //   (one s_add per unit), vector offset = lane * 16, through a ring of RQ_PFU units.  Units past the owned ones
            //   read the next block row, or zeros past the end of the buffer: loaded, never used.
            const unsigned vlane = (unsigned)lane * 16u;
            const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(P.Apack2), 0, NB * KS * 512, 0x00020000);
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
            for (int64_t i = pw; i < gmax; i += 2) {
                // block row of product i + 2 and the two blocks the chain supplies itself (being rewritten / rewritten last)
                int row2 = row + 2; row2 = row2 >= NB ? row2 - NB : row2;
                // ALWAYS two holes, also for the first products of an episode (the chain supplies them from the tile as if the
                // sweep before had just ended): every product of the kernel is summed in the same association, so a restart's
                // values do not depend on where episode boundaries fall (i.e. on the scheduling of the other slots)
                const int h1 = row == 0 ? NB - 1 : row - 1;
                const int h2 = h1 == 0 ? NB - 1 : h1 - 1;
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
                {
                    double *part = part2 + (int)(i & 1) * RQ_NSIMD * 256 + sm * 256;
    #pragma unroll
                    for (int v = 0; v < 4; v++) part[v * 64 + (lane & 15) * 4 + (lane >> 4)] = acc[v];
                }
                rq_sync_write(sy, RQ_PARTS + mw, (int)i + 1, lane);
                row = row2;
            }
    #undef RQ_LDA
        } else {
            // ========================================================================== chain role
            __builtin_amdgcn_s_setprio(3);
            // feasible set of this lane's restart for the episode (from the table the refill step keeps): [-symb, -syma] u [syma, symb]
            const int Un = TC.n[r], Uslow = TC.slow[r];
            const double Ul0 = TC.lo[r], Uh0 = TC.hi[r], Ul1 = TC.lo[16 + r], Uh1 = TC.hi[16 + r];
            const bool two = Un >= 2;
            const double thr = two ? 1e-7 * (Ul1 - Uh0) : 0.0;
            const double syma = two ? Ul1 : 0.0, symb = two ? Uh1 : Uh0;
            ChainState S;
            S.fcur = 0.0; S.upd_counter = cst[0 * 64 + lane]; S.visits = cst[1 * 64 + lane]; S.accepted = cst[2 * 64 + lane];
            S.sweeps = cst[3 * 64 + lane]; S.conv = cst[4 * 64 + lane] != 0; S.status = (int)cst[5 * 64 + lane];
            double fpart = __longlong_as_double(cst[6 * 64 + lane]);
            // lifecycle mode.  The objective of the result is not tracked from an evaluated start value (there is no
            // evaluation pass): `facc` sums x_i ((P0 x)_i + q_i) over the visits since the restart's last move -- when it
            // converges (n visits without a move, qcqp.py:172-176) those are all n coordinates at the FINAL point, i.e.
            // f0(x) - r0 freshly evaluated from the products the sweep computed anyway.  A restart that stops otherwise (sweep
            // limit, gate not passed) takes one FROZEN sweep (no moves, not counted) that sums the same terms.
            bool frz = LIFE && (cst[7 * 64 + lane] & 1) != 0, done = LIFE && (cst[7 * 64 + lane] & 2) != 0;
            double facc = LIFE ? __longlong_as_double(cst[8 * 64 + lane]) : 0.0;
            const bool occupied = sid[r] >= 0;                  // a restart sits in this lane's slot
            const RqOwn cown = rq_own(NB, CS, RQ_NSIMD);
            v2d_ arC[2 * CSU];
            double bqC[4 * CSU];
            double afix[4] = {0.0, 0.0, 0.0, 0.0}, afix2[4] = {0.0, 0.0, 0.0, 0.0};
            v4d_ carry = {0.0, 0.0, 0.0, 0.0};      // the block rewritten last times the fragments of the row after next
            {
                // virtual interval before the episode: the two blocks that a sweep rewrites last (NB - 2, NB - 1) times the
                // fragments of block rows 0 and 1, exactly as the end of a sweep leaves them (carry: NB - 1 x row 1; the
                // chain's plane: NB - 2 and NB - 1 x row 0 + the chain's share of row 0 without those two)
                v4d_ c0 = {0.0, 0.0, 0.0, 0.0}, c1 = {0.0, 0.0, 0.0, 0.0};
                const int bl2 = NB - 2, bl1 = NB - 1;
                double f0a[4], f0b[4], f1b[4], x2[4], x1[4];
    #pragma unroll
                for (int u = 0; u < 4; u++) {
                    f0a[u] = P.Apack[((int64_t)0 * KS + 4 * bl2 + u) * 64 + lane];
                    f0b[u] = P.Apack[((int64_t)0 * KS + 4 * bl1 + u) * 64 + lane];
                    f1b[u] = P.Apack[((int64_t)1 * KS + 4 * bl1 + u) * 64 + lane];
                    x2[u] = Xs[(4 * bl2 + u) * 64 + (lane >> 4) * 16 + (lane & 15)];
                    x1[u] = Xs[(4 * bl1 + u) * 64 + (lane >> 4) * 16 + (lane & 15)];
                }
    #pragma unroll
                for (int u = 0; u < 4; u++) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0a[u], x2[u], c0, 0, 0, 0);   // what `carry` held for row 0
    #pragma unroll
                for (int u = 0; u < 4; u++) {
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(f0b[u], x1[u], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(f1b[u], x1[u], c1, 0, 0, 0);
                }
                carry = c1;
                if (CS > 0) {
                    rq_load_A<CSU>(arC, P.Apack2, KS, cown, lane, 0);
                    rq_load_B<CSU>(bqC, Xs, cown, lane);
                    c0 = rq_product<CSU>(arC, bqC, P.Apack2, KS, cown, lane, rq_slot(cown, bl1), rq_slot(cown, bl2), 1, c0);
                }
    #pragma unroll
                for (int v = 0; v < 4; v++) fixp[v * 64 + (lane & 15) * 4 + (lane >> 4)] = c0[v];
            }
            const double tolv = a.tol;
            int b = 0;
            for (int64_t g = 0; g < gmax; g++) {
                const int bn = (b + 1 == NB) ? 0 : b + 1, bn2 = (bn + 1 == NB) ? 0 : bn + 1;
                const int bprev = (b == 0) ? NB - 1 : b - 1;
                const int cur = (int)(g & 1);
                const double *DU = DU2 + cur * 256, *rtb = rtb2 + cur * 16, *dgb = dg2 + cur * 16, *hqb = hqb2 + cur * 16;
                const double *part = part2 + cur * RQ_NSIMD * 256;
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
                // ---- G + q/2 of the lane's own columns: its own plane, then the three partial tiles, in a fixed order, then q/2
                double gb[4], g0[4], xo[4], xn[4], rto[4], t2o[4], hq4[4];
    #pragma unroll
                for (int v = 0; v < 4; v++) {
                    double s = fixp[v * 64 + lane];
    #pragma unroll
                    for (int w = 0; w < RQ_NSIMD; w++) s += part[w * 256 + v * 64 + lane];
                    hq4[v] = hqb[4 * v + gq];
                    s += hq4[v];
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
                    const double *ap = P.Apack + ((int64_t)bn * KS + 4 * b) * 64 + lane;
                    const double *ap2 = P.Apack + ((int64_t)bn2 * KS + 4 * b) * 64 + lane;
    #pragma unroll
                    for (int u = 0; u < 4; u++) { afix[u] = ap[u * 64]; afix2[u] = ap2[u * 64]; }
                }
                if (b == 0 && !S.conv && S.sweeps >= a.num_iters) {      // sweep limit reached (qcqp.py:160): the restart is done
                    S.conv = true;
                    if (LIFE) { frz = true; facc = 0.0; }                 // ... after one frozen sweep that evaluates its objective
                }
                if (b == 0 && !S.conv) S.sweeps++;
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
                // per RESTART (the four lanes of a quad each looked at their own columns): does the block need the reference's
                // arithmetic?  Only those restarts walk the generic loop; the others commit what the fast path computed, so
                // that a restart's values never depend on which restarts share its tile (cd_phase2_q_kernel drags the whole
                // tile through the generic loop).
                const bool redo = rq_quad_or((act && Un > 0 && (!allfar || Uslow != 0)) ? 1u : 0u) != 0u;
                auto fast_commit = [&]() {
                    if (act) {
                        fpart += fadd;
                        const int accn = __builtin_popcount(mv);
                        const int upd = mv ? (__builtin_clz(mv) - 16) : (int)S.upd_counter + 16;
                        S.accepted += accn;
                        const int over = upd - (int)P.n;
                        S.visits += 16 - (over > 0 ? over : 0);
                        S.upd_counter = upd;
                        if (over >= 0) S.conv = true;
                        if (LIFE) {
                            // the window: visits after the block's last move (all of them if none moved), up to the visit that
                            // completes the n consecutive visits without a move
                            const int cl = mv ? 31 - __builtin_clz(mv) : -1, ce = 15 - (over > 0 ? over : 0);
                            double w = 0.0;
#pragma unroll
                            for (int v = 0; v < 4; v++) {
                                const int c = 4 * v + gq;
                                w = (c > cl && c <= ce) ? __builtin_fma(xn[v], gb[v] + hq4[v], w) : w;
                            }
                            facc = (cl >= 0 ? 0.0 : facc) + w;
                            if (over >= 0) done = true;
                        }
                    } else if (LIFE && frz) {
                        double w = 0.0;
#pragma unroll
                        for (int v = 0; v < 4; v++) w = __builtin_fma(xo[v], gb[v] + hq4[v], w);
                        facc += w;
                    }
#pragma unroll
                    for (int v = 0; v < 4; v++) Xs[(16 * b + 4 * v + gq) * 16 + r] = xn[v];
                };
                if (__builtin_amdgcn_ballot_w64(redo) == 0ull) {
                    fast_commit();
                } else {
                    // ---- generic loop (rare): the reference's arithmetic; G tile (kept from the block's start: the partial
                    // tiles may already be overwritten) rebuilt in the chain's plane, all four lanes of a quad walk their
                    // restart redundantly (same values, benign identical LDS writes)
                    if (!redo) fast_commit();
                    else {
                        S.fcur = rq_quad_sum(fpart);
                        double fa = LIFE ? rq_quad_sum(facc) : 0.0;
                        double *Gsc = fixp;
                        const uint64_t dseed = sseed[r], dfirst = sfirst[r];
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
                            DrawKey dk{dseed, dfirst + (uint64_t)sid[r], (uint32_t)i, (uint32_t)(S.sweeps - 1) | 0x80000000u, 0u};
                            double xnew = xi;
                            int got = S.conv ? 0 : onevar_minimise<MAXC>(t2g, t1, t0, C, dk, &xnew);
                            bool moved;
                            double delta;
                            const bool wasconv = S.conv;
                            chain_commit<MAXC>(S, got, xnew, xi, t2g, t1, t0, a.tol, P.n, moved, delta);
                            if (LIFE && !wasconv) fa = moved ? 0.0 : fa + xi * (Gsc[c * 16 + r] + hq);
                            if (moved) {
                                Xs[i * 16 + r] = xnew;
                                for (int c2 = c + 1; c2 < 16; c2++) Gsc[c2 * 16 + r] += DU[c * 16 + c2] * delta;
                            }
                        }
                        fpart = (gq == 0) ? S.fcur : 0.0;
                        if (LIFE) { facc = (gq == 0) ? fa : 0.0; if (S.conv) done = true; }
                    }
                }
                rq_sync_write(sy, RQ_COMMIT, (int)g + 1, lane);   // block b is in the X tile; its staged operands are free
                if (LIFE && frz && b == NB - 1) { frz = false; done = true; }      // the frozen sweep is complete
                const unsigned long long livem = __builtin_amdgcn_ballot_w64(!S.conv || frz);
                if (livem == 0ull) break;
                if (b == NB - 1) {
                    // sweep boundary: slots whose restart is done (converged, or at the sweep limit) can take a new restart --
                    // end the episode if the queue has one
                    const bool fin = LIFE ? (!occupied || done) : (!occupied || S.conv || S.sweeps >= a.num_iters);
                    const unsigned long long finm = __builtin_amdgcn_ballot_w64(fin);
                    if (finm == ~0ull) break;
                    if (finm != 0ull) {
                        const bool more = qs_load_int(qs_g(a0.b.next)) < (LIFE ? (int)qs_g(lifep)->Rtotal : (int)a0.b.R);
                        if (more) break;
                    }
                }
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
                        acc = rq_product<CSU>(arC, bqC, P.Apack2, KS, cown, lane, us, rq_slot(cown, bprev), bn2, acc);
                    }
    #pragma unroll
                    for (int v = 0; v < 4; v++) fixp[v * 64 + (lane & 15) * 4 + (lane >> 4)] = acc[v];
                }
                b = bn;
            }
            rq_sync_write(sy, RQ_STOP, 1, lane);
            {
                cst[0 * 64 + lane] = S.upd_counter; cst[1 * 64 + lane] = S.visits; cst[2 * 64 + lane] = S.accepted;
                cst[3 * 64 + lane] = S.sweeps; cst[4 * 64 + lane] = S.conv ? 1 : 0; cst[5 * 64 + lane] = S.status;
                cst[6 * 64 + lane] = __double_as_longlong(fpart);
                if (LIFE) { cst[7 * 64 + lane] = (frz ? 1 : 0) | (done ? 2 : 0); cst[8 * 64 + lane] = __double_as_longlong(facc); }
                const double ftot = LIFE ? rq_quad_sum(facc) + P.r0 : rq_quad_sum(fpart);
                const bool fin = LIFE ? (occupied && done) : (occupied && (S.conv || S.sweeps >= a.num_iters));
                if (gq == 0) {
                    sfin[r] = fin ? 1 : 0;
                    if (fin) { ovis[r] = S.visits; oacc[r] = S.accepted; oswp[r] = S.sweeps; ost[r] = S.status; of0[r] = ftot; }
                }
            }
        }

        __syncthreads();
        if (LIFE && tid == 0 && qs_g(lifep)->prof)
            atomicAdd((unsigned long long *)qs_g(lifep)->prof + 5, (unsigned long long)((long long)__builtin_amdgcn_s_memtime() - *(long long *)(ctl + 6)));
        // ================================================================ write out the slots that finished
        {
            // max violation of the final points, same expression as eval_kernel: (p x + q) x + r of the one constraint
            // every coordinate carries (single class, one constraint per coordinate)
            const int e0 = P.cptr[P.krep[0]];
            const double cp = P.cp[e0], cq = P.cq[e0], cr = P.cr[e0];
            const int rel = P.crel[e0];
            const int col = tid & 15, slot = tid >> 4;
            double v = -QM_INF;
            if (sfin[col]) {
                QG double *dst = qs_g(a0.b.X) + ((int64_t)(sid[col] >> 4) * n16) * 16 + (sid[col] & 15);
                for (int64_t i = slot; i < n16; i += 32) {
                    const double x = Xs[i * 16 + col];
                    dst[i * 16] = x;
                    if (i < P.n) {
                        const double f = (cp * x + cq) * x + cr;
                        const double w = (rel == RELOP_EQ) ? fabs(f) : (f > 0.0 ? f : 0.0);
                        v = w > v ? w : v;
                    }
                }
            }
            double *red = part2;                 // 512 doubles of the partial-tile area, free between episodes
            red[tid] = v;
            __syncthreads();
            if (tid < 16 && sfin[tid]) {
                double m = -QM_INF;
                for (int s2 = 0; s2 < 32; s2++) { const double w = red[s2 * 16 + tid]; m = w > m ? w : m; }
                const CdBatchG B = qs_batch(a0.b);
                const int id = sid[tid];
                {
                    B.visits[id] = ovis[tid]; B.accepted[id] = oacc[tid]; B.sweeps[id] = oswp[tid]; B.status[id] = ost[tid];
                    if (B.f0out) B.f0out[id] = of0[tid];
                    if (B.mvout) B.mvout[id] = m;
                    if (LIFE) {
                        QG const CdLife *lf = qs_g(lifep);
                        qs_g(lf->sweeps1)[id] = p1sw[tid]; qs_g(lf->status1)[id] = p1st[tid];
                        qs_g(lf->ran2)[id] = (uint8_t)gatep[tid];
                    }
                }
                sid[tid] = -1; sfin[tid] = 0;
            }
            __syncthreads();
        }
    }
    if (LIFE && tid0 == 0 && qs_g(a0.life)->prof)
        atomicAdd((unsigned long long *)qs_g(a0.life)->prof + 1, (unsigned long long)((long long)__builtin_amdgcn_s_memtime() - life_t0));
}

}  // namespace


size_t cd_queue_lds_bytes(const DevProblem &P) {
    const int NB = (int)P.NB;
    if (P.n % 16 != 0 || NB < 3) return 0;
    size_t bytes = ((size_t)RQ_LDS_COMMON + 8 + 16 * 5 + 8 * 6 + 64 * 10 + 32 + 16 + 8 * 5 + (size_t)P.n16 * 16) * sizeof(double);
    if (bytes < RQ_LDS_MIN + 1024) bytes = RQ_LDS_MIN + 1024;
    return bytes <= 160 * 1024 ? bytes : 0;
}

int cd_queue_launch(const CdQueueArgs &a, int cs, int max_wgs, hipStream_t st) {
    const size_t lds = cd_queue_lds_bytes(a.P);
    if (!lds) return (int)hipErrorInvalidValue;
    const int NB = (int)a.P.NB;
    cs = cs > RQ_CSMAX ? RQ_CSMAX : cs;
    if (cs >= NB) cs = 0;
    cs &= ~1;
    if (NB - cs > RQ_NSIMD * RQ_MAXU) return (int)hipErrorInvalidValue;
    auto k = a.life_on ? (cs == 0 ? cd_phase2_qs_kernel<0, true> : cs == 2 ? cd_phase2_qs_kernel<2, true> : cs == 4 ? cd_phase2_qs_kernel<4, true> : cd_phase2_qs_kernel<6, true>)
                       : (cs == 0 ? cd_phase2_qs_kernel<0, false> : cs == 2 ? cd_phase2_qs_kernel<2, false> : cs == 4 ? cd_phase2_qs_kernel<4, false> : cd_phase2_qs_kernel<6, false>);
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    int64_t wgs = (a.b.R + 15) / 16;     // lifecycle mode: b.R = all restarts of the run
    if (wgs > max_wgs) wgs = max_wgs;
    if (wgs < 1) wgs = 1;
    hipLaunchKernelGGL(k, dim3((unsigned)wgs), dim3(512), lds, st, a);
    return (int)hipGetLastError();
}

}  // namespace qcqpmi
