// cd_wave_kernel -- a restart's whole improve step (suggest(RANDOM), phase 1, gate, phase 2, objective, max violation; the
// lifecycle of cd_queue.h) for the Boolean family with ONE WAVEFRONT PER RESTART and an INCREMENTAL gradient (round 4).
//
// cd_phase2_qs_kernel recomputes G = P0[I_b, :] X on the matrix cores for every block of 16 coordinates of every sweep:
// 2 n^2 flops per restart-sweep whether or not anything moves -- and after the first sweep almost nothing does (n = 1024:
// 266 of the 359 moves of a restart are made in its first sweep, 9.5 sweeps on average).  Here the matrix cores multiply
// ONCE per restart (G = P0 X for the tile of 16 restarts a workgroup owns, at the start of phase 2); from then on a wave
// keeps its restart's point and gradient in registers (coordinate 128 k + 2 l + s in lane l, register pair k) and
//   * decides all 128 coordinates of a window at once (every lane its two, assuming nothing before them moves),
//   * finds the first one that moves with two ballots, commits the visits before it (they cannot move: nothing changed),
//   * makes that move: one row of P0 (8 KB, coalesced 16-byte loads, L2-resident) times the step into the gradient,
//   * decides the rest of the window again.
// The visits, their order, the move rule (qcqp.py:152-178: |x_new - x_i| > tol) and the stopping rule (n consecutive visits
// without a move) are those of the reference; the per-visit arithmetic is that of cd_phase2_q_kernel's fast path (same
// expressions), near-ties and non-standard feasible sets take the reference's arithmetic (onevar_minimise) like there.  What
// differs from the product kernels is rounding only: the gradient is P0 x accumulated move by move (one fma per entry and
// move) instead of summed afresh -- after a few hundred moves 1e-13 relative, against decisions that are discrete (a sign
// flip moves x by 2) or guarded (near-ties).  Cost per restart: one product (2 n^2 flops on the matrix cores, P0 read once per
// 16 restarts) + moves x 2 n flops, instead of sweeps x 2 n^2.
//
// Workgroup = 16 waves = the 16 restarts of tile blockIdx.x of the run's queue (cd_queue.h: CdLife); no slot scheduling: a
// restart's cost is dominated by its moves, which vary little, and the hardware's workgroup queue balances the CUs.
#include "cd_queue.h"

#include "onevar.h"
#include "cd_phase1_sep.h"

namespace qcqpmi {
// (cd_phase2.h, which ChainState / chain_commit / compute_set come with, expects the MFMA building block of kernels.hip to be declared)
typedef double v4d __attribute__((ext_vector_type(4)));
template <typename XPtr>
__device__ inline v4d block_rows_times_X(const double *__restrict__ Ab, XPtr Xs, int kk0, int kk1, int lane, v4d acc) {
    const int xoff = (lane >> 4) * 16 + (lane & 15);
    for (int kk = kk0; kk < kk1; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ab[(int64_t)kk * 64 + lane], Xs[kk * 64 + xoff], acc, 0, 0, 0);
    return acc;
}
}  // namespace qcqpmi

#include "cd_phase2.h"

namespace qcqpmi {
namespace {

#define QG __attribute__((address_space(1)))
template <class T>
__device__ __attribute__((always_inline)) inline QG T *wv_g(T *p) { return (QG T *)p; }

typedef double wv_v2d __attribute__((ext_vector_type(2)));
typedef double wv_v4d __attribute__((ext_vector_type(4)));

__device__ inline double wv_wave_max(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const double w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}
__device__ inline double wv_wave_sum(double v) {      // fixed butterfly: the same association for every restart
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline int wv_wave_min_int(int v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const int w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
    return v;
}
// value of lane `src` (wave-uniform, not a compile-time constant) as a wave-uniform double
__device__ __attribute__((always_inline)) inline double wv_lane(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __attribute__((always_inline)) inline double wv_first(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// the normals of the element pair (elem, elem + 1), elem even: keyed_normal's own expressions (philox.h)
__device__ __attribute__((noinline)) double wv_keyed_normal_pair(uint64_t seed, uint64_t restart, uint64_t elem, double *odd) {
    const U4 o = philox4x32_10((uint32_t)(elem >> 1), (uint32_t)(elem >> 33), 0xA5A50000u, (uint32_t)restart, (uint32_t)seed,
                               (uint32_t)(seed >> 32) ^ (uint32_t)(restart >> 32));
    const double u1 = (((double)(o.x >> 5) * 67108864.0 + (double)(o.y >> 6)) + 0.5) / 9007199254740992.0;
    const double u2 = u53(o.z, o.w);
    const double rad = sqrt(-2.0 * log(u1));
    const double ang = 6.283185307179586476925286766559 * u2;
    *odd = rad * sin(ang);
    return rad * cos(ang);
}
// one phase-1 visit (cd_phase1_sep.h: the moves of cd_phase1_sep_kernel and of the slot-queue kernel bit for bit)
__device__ __attribute__((noinline)) double wv_p1_visit(double p, double q, double r, int relop, int64_t i, double x, double tol,
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
// the reference's arithmetic for one visit (near-ties, feasible sets with infinite end points): uniform over the wave
struct WvGeneric { double xnew, delta, fcur; int moved, conv, status; long long upd, visits, accepted; };
__device__ __attribute__((noinline)) void wv_generic_visit(double t2g, double hq, double gbi, double xi, int Un, double Ul0, double Uh0,
                                                           double Ul1, double Uh1, uint64_t seed, uint64_t restart, uint32_t coord,
                                                           uint32_t sweep_tag, double tol, int64_t n, WvGeneric *io) {
    FeasSet<1> C;
    C.n = Un; C.lo[0] = Ul0; C.hi[0] = Uh0; C.lo[1] = Ul1; C.hi[1] = Uh1;
    const double t1 = 2.0 * ((gbi - hq) - t2g * xi) + (hq + hq);
    const double t0 = io->fcur - xi * (t2g * xi + t1);
    DrawKey dk{seed, restart, coord, sweep_tag, 0u};
    double xnew = xi;
    const int got = onevar_minimise<1>(t2g, t1, t0, C, dk, &xnew);
    ChainState S;
    S.fcur = io->fcur; S.upd_counter = io->upd; S.visits = io->visits; S.accepted = io->accepted; S.sweeps = 0; S.conv = false;
    S.status = io->status;
    bool moved;
    double delta;
    chain_commit<1>(S, got, xnew, xi, t2g, t1, t0, tol, n, moved, delta);
    io->xnew = xnew; io->delta = delta; io->fcur = S.fcur; io->moved = moved ? 1 : 0; io->conv = S.conv ? 1 : 0; io->status = S.status;
    io->upd = S.upd_counter; io->visits = S.visits; io->accepted = S.accepted;
}

constexpr int WV_NONE = 1 << 20;

__device__ __attribute__((always_inline)) inline int wv_ufirst(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __attribute__((always_inline)) inline long long wv_ufirst(long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
}

// NW: windows of 128 coordinates (n <= 128 NW); FULL: n = 128 NW exactly (no guards on the last window)
template <int NW, bool FULL>
__global__ __launch_bounds__(1024) void cd_wave_kernel(CdQueueArgs a0) {
    extern __shared__ double smem[];
    const DevProblem &P = a0.P;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = FULL ? 128 * NW : (int)P.n16;
    const int n = FULL ? 128 * NW : (int)P.n;
    const int NB = (int)P.NB, KS = (int)P.KS;
    // LDS: the tile area holds the restarts' points -- column-wise ([restart][n16], a wave's own column: conflict-free pair reads)
    // except around the product, which needs the B-operand layout [n16][16]; the second area holds one pass of the product's
    // result ([16 restarts][256 rows]) and afterwards the per-coordinate constants of the fast path
    double *Xa = smem;
    double *Xw = smem + (int64_t)wave * n16;        // this wave's column
    double *Gx = smem + (int64_t)n16 * 16;
    QG const CdLife *lf = wv_g(a0.life);
    const bool prof = lf->prof != nullptr && tid == 0;
    const long long t_begin = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
    const int64_t id = (int64_t)blockIdx.x * 16 + wave;
    const bool occ = id < lf->Rtotal;
    const uint64_t pop = occ ? (uint64_t)id / (uint64_t)lf->Rpop : 0, rho = occ ? (uint64_t)id % (uint64_t)lf->Rpop : 0;
    const uint64_t sd = lf->seed + pop * lf->seed_stride;
    const uint64_t gidx = lf->first_index + pop * lf->first_stride + rho;
    const double tolv = a0.tol, viol_tol = lf->viol_tol;
    const int e0 = P.cptr[P.krep[0]];
    const double cp = P.cp[e0], cq = P.cq[e0], cr = P.cr[e0];
    const int rel = P.crel[e0];
    const int nwin = (n + 127) >> 7;                // windows in use

    // ================================================================ the restart's start: normals / uploaded point, phase 1
    int sweeps1 = 0, status1 = 0;
    if (occ && lf->generate) {
        for (int k = 0; k < nwin; k++) {
            const int i = 128 * k + 2 * lane;           // n is a multiple of 16: a pair never straddles n
            wv_v2d x2 = {0.0, 0.0};
            if (i < n) {
                double xo;
                x2[0] = wv_keyed_normal_pair(sd, gidx, (uint64_t)i, &xo);
                x2[1] = xo;
            }
            if (i < n16) *(wv_v2d *)(Xw + i) = x2;
        }
    } else if (occ) {
        QG const double *src = wv_g(a0.b.X) + ((id >> 4) * n16) * 16 + (id & 15);
        for (int k = 0; k < nwin; k++) {
            const int i = 128 * k + 2 * lane;
            wv_v2d x2 = {0.0, 0.0};
            if (i < n) { x2[0] = src[(int64_t)i * 16]; x2[1] = src[(int64_t)(i + 1) * 16]; }
            if (i < n16) *(wv_v2d *)(Xw + i) = x2;
        }
    } else {
        for (int k = 0; k < nwin; k++) {
            const int i = 128 * k + 2 * lane;
            if (i < n16) *(wv_v2d *)(Xw + i) = wv_v2d{0.0, 0.0};
        }
    }
    if (occ && lf->phase1) {
        for (int64_t t = 0; t < a0.num_iters; t++) {
            double vmax = -QM_INF;
            int upd = 0, st = 0;
            for (int k = 0; k < nwin; k++) {
                const int i = 128 * k + 2 * lane;
                if (i < n) {
                    wv_v2d x2 = *(wv_v2d *)(Xw + i);
                    int fl0, fl1;
                    double va0, va1;
                    const double y0 = wv_p1_visit(cp, cq, cr, rel, i, x2[0], tolv, viol_tol, sd, gidx, t, &fl0, &va0);
                    const double y1 = wv_p1_visit(cp, cq, cr, rel, i + 1, x2[1], tolv, viol_tol, sd, gidx, t, &fl1, &va1);
                    if (fl0 >> 8) st = -(fl0 >> 8);
                    if (fl1 >> 8) st = -(fl1 >> 8);
                    if (fl0 & 1) { x2[0] = y0; upd = 1; }
                    if (fl1 & 1) { x2[1] = y1; upd = 1; }
                    if ((fl0 | fl1) & 1) *(wv_v2d *)(Xw + i) = x2;
                    vmax = va0 > vmax ? va0 : vmax;
                    vmax = va1 > vmax ? va1 : vmax;
                }
            }
            vmax = wv_first(wv_wave_max(vmax));
            st = wv_ufirst(wv_wave_min_int(st));
            sweeps1++;
            if (st) status1 = st;
            // done when feasible enough (qcqp.py:111); a sweep without any update is a fixed point of the map
            if (vmax < viol_tol || __builtin_amdgcn_ballot_w64(upd != 0) == 0ull) break;
        }
    }
    // the point into registers; its max violation = the slack of phase 2 (qcqp.py:157), the gate (qcqp.py:189), the feasible
    // set of a coordinate at that slack
    double gr[2 * NW];
    wv_v2d xk[NW];
    double mvx = -QM_INF;
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const int i = 128 * k + 2 * lane;
        xk[k] = (i < n16) ? *(wv_v2d *)(Xw + i) : wv_v2d{0.0, 0.0};
        gr[2 * k] = 0.0; gr[2 * k + 1] = 0.0;
        if (i < n) {
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const double f = (cp * xk[k][s] + cq) * xk[k][s] + cr;
                const double w = (rel == RELOP_EQ) ? fabs(f) : (f > 0.0 ? f : 0.0);
                mvx = w > mvx ? w : mvx;
            }
        }
    }
    mvx = wv_first(wv_wave_max(mvx));
    const bool gate = occ && mvx < viol_tol && status1 == 0;
    int Un, Uslow;
    double Ul0, Uh0, Ul1, Uh1;
    {
        FeasSet<1> C;
        compute_set<1>(P, P.krep[0], occ ? mvx : 0.0, C);
        Un = wv_ufirst(C.n);
        Ul0 = wv_first(C.lo[0]); Uh0 = wv_first(C.hi[0]); Ul1 = wv_first(C.lo[1]); Uh1 = wv_first(C.hi[1]);
        bool inf = false;
#pragma unroll
        for (int j = 0; j <= 1; j++)
            if (j < C.n && (__builtin_isinf(C.lo[j]) || __builtin_isinf(C.hi[j]))) inf = true;
        Uslow = wv_ufirst(inf ? 1 : 0);
    }
    const long long t_built = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;

    // ================================================================ G = P0 X for the tile on the matrix cores
    __syncthreads();                                // every wave holds its column in registers
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const int i = 128 * k + 2 * lane;
        if (i < n16) { Xa[(int64_t)i * 16 + wave] = xk[k][0]; Xa[(int64_t)(i + 1) * 16 + wave] = xk[k][1]; }
    }
    __syncthreads();
    {
        const int xoff = (lane >> 4) * 16 + (lane & 15);
        const int KS2 = KS >> 1;
        QG const wv_v2d *Ap2 = (QG const wv_v2d *)wv_g(P.Apack2);
#pragma unroll
        for (int p = 0; p < (NW + 1) / 2; p++) {
            const int rb = 16 * p + wave;
            if (rb < NB) {
                wv_v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
                QG const wv_v2d *ab = Ap2 + ((int64_t)rb * KS2) * 64 + lane;
                for (int kk2 = 0; kk2 < KS2; kk2 += 4) {        // KS2 = n16 / 8 is even; the tail is guarded
                    wv_v2d a2[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) a2[u] = (kk2 + u < KS2) ? ab[(int64_t)(kk2 + u) * 64] : wv_v2d{0.0, 0.0};
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (kk2 + u < KS2) {
                            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[u][0], Xa[(2 * (kk2 + u)) * 64 + xoff], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[u][1], Xa[(2 * (kk2 + u) + 1) * 64 + xoff], acc1, 0, 0, 0);
                        }
                    }
                }
                acc0 = acc0 + acc1;
                // acc[v] of lane l: coordinate 16 rb + 4 v + (l >> 4) of restart l & 15
#pragma unroll
                for (int v = 0; v < 4; v++) Gx[(lane & 15) * 256 + 16 * wave + 4 * v + (lane >> 4)] = acc0[v];
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                const int k = 2 * p + kk;
                if (k < NW) {
                    const int i = 128 * k + 2 * lane;
                    if (i < n16) {
                        // the restart carries G + q / 2 (what the decision reads), like the product kernels' tile sum
                        const wv_v2d g2 = *(const wv_v2d *)(Gx + wave * 256 + 128 * kk + 2 * lane);
                        const wv_v2d q2 = *(QG const wv_v2d *)(wv_g(P.q0) + i);
                        gr[2 * k] = g2[0] + 0.5 * q2[0]; gr[2 * k + 1] = g2[1] + 0.5 * q2[1];
                    }
                }
            }
            __syncthreads();
        }
    }
    // the points back into their columns; per-coordinate constants of the fast path: P0[i,i], q0[i] / 2, 1 / P0[i,i]
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const int i = 128 * k + 2 * lane;
        if (i < n16) *(wv_v2d *)(Xw + i) = xk[k];
    }
    double *cD = Gx, *cH = Gx + n16, *cR = Gx + 2 * n16;
    for (int i = tid; i < n16; i += 1024) {
        const double rc = P.rcp2d[i];
        cD[i] = P.P0[(int64_t)i * n16 + i]; cH[i] = 0.5 * P.q0[i]; cR[i] = rc + rc;
    }
    __syncthreads();
    const long long t_mult = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;

    // ================================================================ phase 2 (qcqp.py:152-178), one wave per restart
    int upd = 0, sweeps = 0, status = 0;
    long long visits = 0, accepted = 0;
    double fpart = 0.0;                             // objective relative to the start of phase 2 (feeds the reference-arithmetic visits)
    if (gate) {
        const bool two = Un >= 2;
        const double thr = two ? 1e-7 * (Ul1 - Uh0) : 0.0;
        const double syma = two ? Ul1 : 0.0, symb = two ? Uh1 : Uh0;
        const double tole = Un > 0 ? tolv : QM_INF;  // an empty feasible set never moves
        const bool slowset = Un > 0 && Uslow != 0;
        const bool anynear = Un > 0;
        const int itmax = a0.num_iters > 0x7fffffff ? 0x7fffffff : (int)a0.num_iters;
        // rows of P0 as 16-byte pairs; a lane past the row's end re-reads the last pair (its entries are never used)
        const int lo2 = (FULL || 2 * lane < n16) ? lane : (n16 >> 1) - 1;
        bool conv = false;
        while (!conv && sweeps < itmax) {
            sweeps++;
            for (int k = 0; k < nwin && !conv; k++) {
                const int base = 128 * k;
                const int wlen = FULL ? 128 : ((n - base) < 128 ? (n - base) : 128);
                const int ci = (FULL || base + 2 * lane < n16) ? base + 2 * lane : n16 - 2;
                const wv_v2d r2 = *(const wv_v2d *)(cR + ci);
                wv_v2d x2 = *(const wv_v2d *)(Xw + ci);
                // the window's own entries of G + q / 2: a copy that takes every update the registers take
                double gw0 = gr[0], gw1 = gr[1];
#pragma unroll
                for (int j = 1; j < NW; j++) { gw0 = (k == j) ? gr[2 * j] : gw0; gw1 = (k == j) ? gr[2 * j + 1] : gw1; }
                const bool in0 = 2 * lane < wlen, in1 = 2 * lane + 1 < wlen;
                int start = 0;
                for (;;) {
                    // every lane decides its two coordinates as if nothing before them in the window moved
                    const double xv0 = __builtin_fma(-gw0, r2[0], x2[0]), xv1 = __builtin_fma(-gw1, r2[1], x2[1]);
                    const double pick0 = __builtin_copysign(fmin(fmax(fabs(xv0), syma), symb), xv0);
                    const double pick1 = __builtin_copysign(fmin(fmax(fabs(xv1), syma), symb), xv1);
                    const double dl0 = pick0 - x2[0], dl1 = pick1 - x2[1];
                    const bool near0 = anynear && (!(fabs(xv0) > thr) || slowset), near1 = anynear && (!(fabs(xv1) > thr) || slowset);
                    const bool live0 = in0 && 2 * lane >= start, live1 = in1 && 2 * lane + 1 >= start;
                    const unsigned long long be = __builtin_amdgcn_ballot_w64(live0 && (fabs(dl0) > tole || near0));
                    const unsigned long long bo = __builtin_amdgcn_ballot_w64(live1 && (fabs(dl1) > tole || near1));
                    const int fe = be ? 2 * (int)__builtin_ctzll(be) : WV_NONE, fo = bo ? 2 * (int)__builtin_ctzll(bo) + 1 : WV_NONE;
                    const int first = fe < fo ? fe : fo;
                    const int stop = first == WV_NONE ? wlen : first;
                    // the visits before it do not move (qcqp.py:172-176: n of them in a row end the restart)
                    const int cnt = stop - start;
                    if (upd + cnt >= n) { visits += n - upd; upd = n; conv = true; break; }
                    upd += cnt; visits += cnt;
                    if (first == WV_NONE) break;
                    const int L = first >> 1;
                    const bool s = (first & 1) != 0;
                    // the row of P0 the move will need (symmetric: row = column), issued before anything else
                    QG const wv_v2d *row = (QG const wv_v2d *)(wv_g(P.P0) + (int64_t)(base + first) * n16) + lo2;
                    wv_v2d rv[NW];
#pragma unroll
                    for (int j = 0; j < NW; j++) rv[j] = (FULL || 128 * j < n16) ? row[(FULL || 128 * j + 128 <= n16) ? 64 * j : 0] : wv_v2d{0.0, 0.0};
                    const wv_v2d rw = row[(FULL || base + 128 <= n16) ? 64 * k : 0];
                    const double gbi = wv_lane(s ? gw1 : gw0, L), di = cD[base + first];
                    const unsigned long long bn = __builtin_amdgcn_ballot_w64(s ? near1 : near0);
                    double delta, xnew;
                    if (__builtin_expect(!((bn >> L) & 1ull), 1)) {
                        xnew = wv_lane(s ? pick1 : pick0, L);
                        delta = wv_lane(s ? dl1 : dl0, L);
                        // f(x + d e_i) - f(x) = d (t2 d + 2 g),  g = G_i + q_i / 2 (contains P_ii x_i)
                        fpart += __builtin_fma(delta, __builtin_fma(di, delta, gbi + gbi), 0.0);
                        visits++; accepted++; upd = 0;
                    } else {
                        WvGeneric io;
                        io.fcur = fpart; io.upd = upd; io.visits = visits; io.accepted = accepted; io.status = status;
                        wv_generic_visit(di, cH[base + first], gbi, wv_lane(s ? x2[1] : x2[0], L), Un, Ul0, Uh0, Ul1, Uh1, sd, gidx,
                                         (uint32_t)(base + first), (uint32_t)(sweeps - 1) | 0x80000000u, tolv, (int64_t)n, &io);
                        fpart = wv_first(io.fcur); upd = wv_ufirst((int)io.upd); visits = wv_ufirst(io.visits); accepted = wv_ufirst(io.accepted);
                        status = wv_ufirst(io.status);
                        delta = wv_ufirst(io.moved) ? wv_first(io.delta) : 0.0;
                        xnew = wv_ufirst(io.moved) ? wv_first(io.xnew) : wv_lane(s ? x2[1] : x2[0], L);
                        if (wv_ufirst(io.conv)) conv = true;
                    }
                    start = first + 1;
                    if (lane == L) { if (s) x2[1] = xnew; else x2[0] = xnew; *(wv_v2d *)(Xw + ci) = x2; }
                    // the gradient follows the move (a visit that did not move: step 0)
                    gw0 = __builtin_fma(rw[0], delta, gw0); gw1 = __builtin_fma(rw[1], delta, gw1);
#pragma unroll
                    for (int j = 0; j < NW; j++) {
                        gr[2 * j] = __builtin_fma(rv[j][0], delta, gr[2 * j]);
                        gr[2 * j + 1] = __builtin_fma(rv[j][1], delta, gr[2 * j + 1]);
                    }
                    if (conv) break;
                }
            }
        }
    }
    // ================================================================ objective and max violation of the final point
    // f0 = sum_i x_i ((P0 x)_i + q_i) + r0 from the gradient the restart carries; same expression per term as the slot-queue
    // kernel's window sum
    double fsum = 0.0, mvf = -QM_INF;
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const int i = 128 * k + 2 * lane;
        xk[k] = (i < n16) ? *(wv_v2d *)(Xw + i) : wv_v2d{0.0, 0.0};
        if (i < n) {
            const wv_v2d h2 = *(const wv_v2d *)(cH + i);
#pragma unroll
            for (int s = 0; s < 2; s++) {
                fsum = __builtin_fma(xk[k][s], gr[2 * k + s] + h2[s], fsum);
                const double f = (cp * xk[k][s] + cq) * xk[k][s] + cr;
                const double w = (rel == RELOP_EQ) ? fabs(f) : (f > 0.0 ? f : 0.0);
                mvf = w > mvf ? w : mvf;
            }
        }
    }
    fsum = wv_wave_sum(fsum) + P.r0;
    mvf = wv_wave_max(mvf);
    if (occ && lane == 0) {
        wv_g(a0.b.visits)[id] = visits; wv_g(a0.b.accepted)[id] = accepted; wv_g(a0.b.sweeps)[id] = sweeps; wv_g(a0.b.status)[id] = status;
        if (a0.b.f0out) wv_g(a0.b.f0out)[id] = fsum;
        if (a0.b.mvout) wv_g(a0.b.mvout)[id] = mvf;
        wv_g(lf->sweeps1)[id] = sweeps1; wv_g(lf->status1)[id] = status1; wv_g(lf->ran2)[id] = (uint8_t)(gate ? 1 : 0);
    }
    const long long t_done = prof ? (long long)__builtin_amdgcn_s_memtime() : 0;
    __syncthreads();                                // every wave holds its final column in registers
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const int i = 128 * k + 2 * lane;
        if (i < n16) { Xa[(int64_t)i * 16 + wave] = xk[k][0]; Xa[(int64_t)(i + 1) * 16 + wave] = xk[k][1]; }
    }
    __syncthreads();
    {
        QG wv_v2d *dst = (QG wv_v2d *)(wv_g(a0.b.X) + (int64_t)blockIdx.x * n16 * 16);
        const wv_v2d *srcl = (const wv_v2d *)Xa;
        for (int e = tid; e < n16 * 8; e += 1024) dst[e] = srcl[e];
    }
    if (prof) {
        unsigned long long *pr = (unsigned long long *)lf->prof;
        const long long t_end = (long long)__builtin_amdgcn_s_memtime();
        atomicAdd(pr + 0, (unsigned long long)(t_built - t_begin));      // start of the restarts (normals, phase 1, gate)
        atomicAdd(pr + 1, (unsigned long long)(t_end - t_begin));        // the workgroup
        atomicAdd(pr + 2, 1ull);
        atomicAdd(pr + 3, 16ull);
        atomicAdd(pr + 4, (unsigned long long)(t_mult - t_built));       // the product on the matrix cores
        atomicAdd(pr + 5, (unsigned long long)(t_done - t_mult));        // phase 2 of wave 0's restart
    }
}

}  // namespace

size_t cd_wave_lds_bytes(const DevProblem &P) {
    if (P.n % 16 != 0 || P.n > 1024 || P.NB < 1 || (P.KS & 1)) return 0;
    const size_t bytes = ((size_t)P.n16 * 16 + 16 * 256) * sizeof(double);
    // (the constants take 3 n16 doubles of the tile area: n16 * 16 >= 3 * n16 always)
    return bytes <= 160 * 1024 ? bytes : 0;
}

int cd_wave_launch(const CdQueueArgs &a, hipStream_t st) {
    const size_t lds = cd_wave_lds_bytes(a.P);
    if (!lds) return (int)hipErrorInvalidValue;
    const int nw = (int)((a.P.n + 127) / 128);
    const bool full = a.P.n == a.P.n16 && (a.P.n == 128 || a.P.n == 256 || a.P.n == 512 || a.P.n == 1024);
    auto k = nw <= 1 ? (full ? cd_wave_kernel<1, true> : cd_wave_kernel<1, false>) : nw <= 2 ? (full ? cd_wave_kernel<2, true> : cd_wave_kernel<2, false>)
           : nw <= 4 ? (full ? cd_wave_kernel<4, true> : cd_wave_kernel<4, false>) : (full ? cd_wave_kernel<8, true> : cd_wave_kernel<8, false>);
    hipError_t e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const int64_t wgs = (a.b.R + 15) / 16;
    hipLaunchKernelGGL(k, dim3((unsigned)wgs), dim3(1024), lds, st, a);
    return (int)hipGetLastError();
}

}  // namespace qcqpmi
