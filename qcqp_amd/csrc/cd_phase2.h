// Coordinate descent phase 2 (qcqp.py:152-178) as blocked Gauss-Seidel on the fp64 matrix cores.
//
// One workgroup (4 waves) owns a tile of 16 restarts.  For a block I_b of 16 coordinates
//   1. MFMA:   G = P0[I_b, :] X        16 x n16 by n16 x 16 on v_mfma_f64_16x16x4_f64, K split
//                                       over the 4 waves, partial tiles summed through LDS in a
//                                       fixed order (deterministic);
//   2. stage:  thread (c, r) sums the partials and lays out, for coordinate c of restart r,
//              everything the sequential part needs at STATIC LDS addresses;
//   3. chain:  lanes 0..15 of wave 0 (lane = restart) visit the 16 coordinates in order; each
//              accepted move is folded into the remaining rows of the block through the 16 x 16
//              diagonal block of P0 (exact Gauss-Seidel, not Jacobi).
// The one-variable feasible sets of separable constraints depend only on the constraint
// coefficients and on the restart's slack, both fixed during phase 2 (qcqp.py:157,167): they are
// computed ONCE per kernel for every "constraint class" (coordinates whose constraint lists are
// bit-identical share a class) when the class table fits LDS, per block otherwise.
//
// The per-coordinate step is straight-line code (finite end points, no ties, a non-degenerate
// scalar objective).  Anything else -- infinite end points, exact ties between end points
// (np.random.choice), the zero-objective branch (np.random.uniform) -- is detected per wave and
// the rest of the block is finished by a compact generic loop built on onevar_minimise.
#pragma once
#include "kernels.h"
#include "onevar.h"

namespace qcqpmi {

typedef double v4d_ __attribute__((ext_vector_type(4)));

template <int MAXC>
struct SetTable {
    int *n;        // [slots] number of intervals
    int *slow;     // [slots] 1 if an end point is infinite (generic path needed)
    double *lo;    // [(MAXC+1)][slots]
    double *hi;    // [(MAXC+1)][slots]
    int slots;
};

template <int MAXC>
__device__ inline void store_set(const SetTable<MAXC> &T, int slot, const FeasSet<MAXC> &C) {
    bool inf = false;
#pragma unroll
    for (int j = 0; j <= MAXC; j++) {
        T.lo[j * T.slots + slot] = C.lo[j];
        T.hi[j * T.slots + slot] = C.hi[j];
        if (j < C.n && (__builtin_isinf(C.lo[j]) || __builtin_isinf(C.hi[j]))) inf = true;
    }
    T.n[slot] = C.n;
    T.slow[slot] = inf ? 1 : 0;
}

template <int MAXC>
__device__ inline void compute_set(const DevProblem &P, int list, double slack, FeasSet<MAXC> &C) {
    const int e0 = P.cptr[list], mf = P.cptr[list + 1] - e0;
    double cp[MAXC], cq[MAXC], cr[MAXC];
    int crel[MAXC];
#pragma unroll
    for (int k = 0; k < MAXC; k++) {
        bool ok = k < mf;
        cp[k] = ok ? P.cp[e0 + k] : 0.0; cq[k] = ok ? P.cq[e0 + k] : 0.0;
        cr[k] = ok ? P.cr[e0 + k] : 0.0; crel[k] = ok ? P.crel[e0 + k] : RELOP_LE;
    }
    if (mf == 1) feasible_set_single<MAXC>(cp[0], cq[0], cr[0], crel[0], slack, C);
    else feasible_set<MAXC>(cp, cq, cr, crel, mf, slack, C);
}

// near-IEEE quotient num/den from a precomputed reciprocal: one Newton correction in fma
// arithmetic (within 1 ulp of the correctly rounded quotient the reference computes).
__device__ inline double div_by_rcp(double num, double den, double rcp) {
    double q = num * rcp;
    double r = __builtin_fma(-q, den, num);
    return __builtin_fma(r, rcp, q);
}

// wave-uniform broadcast of a double held by lane `src` (static lane index)
__device__ inline double readlane_d(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// feasible set of one (coordinate, restart) as the fast path consumes it (<= 2 intervals)
struct StepTab {
    double l0, h0, l1, h1, mid, thr;
    int n, slow;
};

// per-restart state of the sequential part
struct ChainState {
    double fcur;
    int64_t upd_counter, visits, accepted, sweeps;
    bool conv;
    int status;
};

template <int MAXC>
__device__ inline void chain_commit(ChainState &S, int got, double xn, double xi, double t2,
                                    double t1, double t0, double tol, int64_t n, bool &moved,
                                    double &delta) {
    moved = false;
    delta = 0.0;
    if (S.conv) return;
    S.visits++;
    if (got < 0) { S.status = got; S.conv = true; return; }
    if (got && fabs(xn - xi) > tol) {
        delta = xn - xi;
        moved = true;
        S.fcur = t0 + xn * (t2 * xn + t1);
        S.upd_counter = 0;
        S.accepted++;
    } else {
        S.upd_counter++;
        if (S.upd_counter == n) S.conv = true;
    }
}

// FAST: 0 = any objective curvature / up to MAXC+1 intervals; 1 = every P0[i,i] > 0 and MAXC == 1;
//       2 = every P0[i,i] == 0 and MAXC == 1.
// UNI : every coordinate has the same constraint class (K == 1): the feasible set of a restart is
//       kept in registers for the whole kernel.
template <int MAXC, bool XLDS, bool CLS, int FAST, bool UNI>
__global__ __launch_bounds__(256) void cd_phase2_kernel(CdArgs a, const double *__restrict__ Apack,
                                                        const double *__restrict__ P0,
                                                        const double *__restrict__ q0,
                                                        const double *__restrict__ rcp2d,
                                                        const int *__restrict__ cls) {
    extern __shared__ double smem[];
    const DevProblem &P = a.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tile = blockIdx.x;
    const int64_t n16 = P.n16;
    double *Xg = a.X + tile * n16 * 16;
    // ---- dynamic LDS carve-up
    double *sp = smem;
    double *Xl = sp; if (XLDS) sp += n16 * 16;
    double *part = sp; sp += 4 * 256;
    double *G = sp; sp += 256;
    double *Dblk = sp; sp += 256;
    double *slk = sp; sp += 16;
    double *q0b = sp; sp += 16;
    double *rcpb = sp; sp += 16;
    double *midb = sp; sp += 256;   // FAST: midpoint of the gap between the two intervals (+inf if < 2)
    double *thrb = sp; sp += 256;   // FAST: near-tie half-width around midb
    // block table: feasible set of (coordinate c, restart r) at static offsets
    SetTable<MAXC> TB;
    TB.slots = 256;
    TB.lo = sp; sp += (MAXC + 1) * 256;
    TB.hi = sp; sp += (MAXC + 1) * 256;
    TB.n = (int *)sp; sp += 128;
    TB.slow = (int *)sp; sp += 128;
    // class table (CLS only)
    SetTable<MAXC> TC;
    TC.slots = CLS ? P.K * 16 : 0;
    TC.lo = sp; sp += (MAXC + 1) * TC.slots;
    TC.hi = sp; sp += (MAXC + 1) * TC.slots;
    TC.n = (int *)sp; sp += (TC.slots + 1) / 2;
    TC.slow = (int *)sp; sp += (TC.slots + 1) / 2;
    int *done = (int *)sp;

    double *Xs = XLDS ? Xl : Xg;
    if (XLDS)
        for (int64_t idx = tid; idx < n16 * 16; idx += 256) Xl[idx] = Xg[idx];
    if (tid < 16) {
        int64_t g = tile * 16 + tid;
        slk[tid] = (g < a.R) ? a.slack[g] : 0.0;
    }
    if (tid == 0) *done = 0;
    __syncthreads();
    if (CLS) {  // feasible sets of every (class, restart) once per kernel
        for (int s = tid; s < P.K * 16; s += 256) {
            FeasSet<MAXC> C;
            compute_set<MAXC>(P, P.krep[s >> 4], slk[s & 15], C);
            store_set<MAXC>(TC, s, C);
        }
    }
    const int r = lane & 15;
    StepTab U;
    U.l0 = U.h0 = U.l1 = U.h1 = U.mid = U.thr = 0.0; U.n = 0; U.slow = 0;
    if (UNI) {
        __syncthreads();
        U.l0 = TC.lo[r]; U.h0 = TC.hi[r]; U.l1 = TC.lo[TC.slots + r]; U.h1 = TC.hi[TC.slots + r];
        U.n = TC.n[r]; U.slow = TC.slow[r];
        const bool two = U.n >= 2;
        U.mid = two ? 0.5 * (U.h0 + U.l1) : QM_INF;
        U.thr = two ? 1e-7 * (U.l1 - U.h0) : 0.0;
    }
    // ---- per-restart chain state (wave 0, lanes 0..15)
    const int64_t gr = tile * 16 + r;
    ChainState S;
    S.fcur = 0.0; S.upd_counter = 0; S.visits = 0; S.accepted = 0; S.sweeps = 0;
    S.conv = true; S.status = 0;
    if (wave == 0 && lane < 16 && gr < a.R) {
        S.conv = a.flag[gr] ? false : true;
        S.fcur = a.f0cur[gr];
    }
    __syncthreads();

    const int kper = (int)(P.KS / 4);  // KS = n16/4 is a multiple of 4
    bool all_done = false;
    long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = 0;
#define PROF_TICK(slot)                                                    \
    if (a.prof) { long long now_ = (long long)__builtin_amdgcn_s_memtime(); pc[slot] += now_ - tp; tp = now_; }
    if (a.prof) tp = (long long)__builtin_amdgcn_s_memtime();

    for (int64_t t = 0; t < a.num_iters && !all_done; t++) {
        if (wave == 0 && lane < 16 && !S.conv) S.sweeps++;
        for (int64_t b = 0; b < P.NB; b++) {
            // ---- 1. G partials on the matrix cores
            v4d_ acc = {0.0, 0.0, 0.0, 0.0};
            acc = block_rows_times_X(Apack + b * P.KS * 64, Xs, wave * kper, (wave + 1) * kper, lane, acc);
#pragma unroll
            for (int v = 0; v < 4; v++)
                part[wave * 256 + ((lane >> 4) + 4 * v) * 16 + (lane & 15)] = acc[v];
            PROF_TICK(0)
            __syncthreads();
            // ---- 2. thread (c, r): sum the partial tiles; stage everything the chain reads
            {
                const int c = tid >> 4;
                const int64_t i = 16 * b + c;
                G[tid] = (part[tid] + part[256 + tid]) + (part[512 + tid] + part[768 + tid]);
                Dblk[tid] = P0[i * n16 + 16 * b + (tid & 15)];
                if ((tid & 15) == 0) { q0b[c] = q0[i]; rcpb[c] = rcp2d[i]; }
                if (CLS) {
                    const int s = cls[i] * 16 + (tid & 15);
                    TB.n[tid] = (i < P.n) ? TC.n[s] : 0;
                    TB.slow[tid] = TC.slow[s];
#pragma unroll
                    for (int j = 0; j <= MAXC; j++) {
                        TB.lo[j * 256 + tid] = TC.lo[j * TC.slots + s];
                        TB.hi[j * 256 + tid] = TC.hi[j * TC.slots + s];
                    }
                } else {
                    FeasSet<MAXC> C;
                    C.n = 0;
#pragma unroll
                    for (int j = 0; j <= MAXC; j++) { C.lo[j] = 0.0; C.hi[j] = 0.0; }
                    if (i < P.n) compute_set<MAXC>(P, (int)i, slk[tid & 15], C);
                    store_set<MAXC>(TB, tid, C);
                }
                if (FAST != 0) {
                    const bool two = TB.n[tid] >= 2;
                    const double h0 = TB.hi[tid], l1 = TB.lo[256 + tid];
                    midb[tid] = two ? 0.5 * (h0 + l1) : QM_INF;
                    thrb[tid] = two ? 1e-7 * (l1 - h0) : 0.0;
                }
            }
            PROF_TICK(1)
            __syncthreads();
            PROF_TICK(2)
            // ---- 3. sequential part: lane = restart
            if (wave == 0) {
                if (lane < 16) {
                    const int cmax = (P.n - 16 * b) < 16 ? (int)(P.n - 16 * b) : 16;  // uniform
                    double xb[16], gb[16];
#pragma unroll
                    for (int c = 0; c < 16; c++) {
                        xb[c] = Xs[(16 * b + c) * 16 + r];
                        gb[c] = G[c * 16 + r];
                    }
                    // ---- straight-line fast path.  Exact-arithmetic restatement of the scalar
                    // minimiser: for a convex one-variable objective the best feasible point is
                    // the projection of the vertex onto the nearest interval; for a concave or
                    // linear one it is one of the two extreme end points.  The reference decides
                    // by comparing ROUNDED objective values, so every decision that is close to a
                    // tie (or touches +-inf, or has a vanishing objective) raises `redo` and the
                    // whole block is recomputed by the generic loop below, which follows the
                    // reference's arithmetic literally.
                    const bool act0 = !S.conv;
                    int upd = (int)S.upd_counter, vis = 0, accn = 0;
                    bool conv = S.conv, redo = false;
                    double fcur = S.fcur;
                    const int nlim = (int)P.n;
                    // the 16 x 16 diagonal block of P0 lives in registers, row c spread over lanes
                    // (lane l holds D[c][l]); uniform multipliers come out by v_readlane.
                    double Dreg[16];
#pragma unroll
                    for (int c = 0; c < 16; c++) Dreg[c] = Dblk[c * 16 + r];
                    const double q0reg = q0b[r], rcpreg = rcpb[r];
                    StepTab tb[16];
                    if (FAST != 0 && !UNI) {
#pragma unroll
                        for (int c = 0; c < 16; c++) {
                            const int slot = c * 16 + r;
                            tb[c].l0 = TB.lo[slot]; tb[c].h0 = TB.hi[slot];
                            tb[c].l1 = TB.lo[256 + slot]; tb[c].h1 = TB.hi[256 + slot];
                            tb[c].mid = midb[slot]; tb[c].thr = thrb[slot];
                            tb[c].n = TB.n[slot]; tb[c].slow = TB.slow[slot];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 16; c++) {
                        if (c >= cmax) continue;   // wave-uniform
                        const int slot = c * 16 + r;
                        const StepTab &W = UNI ? U : tb[c];
                        const int nC = (FAST != 0) ? W.n : TB.n[slot];
                        const int slowset = (FAST != 0) ? W.slow : TB.slow[slot];
                        const double t2 = readlane_d(Dreg[c], c);    // wave-uniform
                        const double rc = readlane_d(rcpreg, c);     // 1 / (2 t2), 0 if t2 == 0
                        const double xi = xb[c];
                        const double t1 = 2.0 * (gb[c] - t2 * xi) + readlane_d(q0reg, c);
                        double pick;
                        bool nr;
                        if (FAST == 1) {
                            // convex, <= 2 intervals: project the vertex on the interval on its side
                            // of the gap's midpoint
                            const double xv = div_by_rcp(-t1, 2.0 * t2, rc);
                            const double p0 = fmin(fmax(xv, W.l0), W.h0);
                            const double p1 = fmin(fmax(xv, W.l1), W.h1);
                            pick = (xv > W.mid) ? p1 : p0;
                            nr = fabs(xv - W.mid) <= W.thr;
                        } else if (FAST == 2) {
                            // linear: the extreme end point against the slope
                            const double L = W.l0;
                            const double H = (nC >= 2) ? W.h1 : W.h0;
                            pick = (t1 > 0.0) ? L : H;
                            nr = fabs(t1) * (fabs(L) + fabs(H)) <= 1e-9 * fabs(fcur) + 1e-300;
                        } else {
                            const double xv = div_by_rcp(-t1, 2.0 * t2, rc);  // vertex (unused if t2 == 0)
                            // convex: nearest projection
                            double bestd = QM_INF, pickc = xi, L = TB.lo[slot], H = TB.hi[slot];
                            bool near = false;
#pragma unroll
                            for (int j = 0; j <= MAXC; j++) {
                                const double lo = TB.lo[j * 256 + slot], hi = TB.hi[j * 256 + slot];
                                const bool v = j < nC;
                                const double cj = fmin(fmax(xv, lo), hi);
                                const double d = fabs(cj - xv);
                                near = near || (v && bestd < QM_INF && fabs(d - bestd) <= 1e-7 * (d + bestd));
                                if (v && d < bestd) { bestd = d; pickc = cj; }
                                if (v) H = hi;
                            }
                            // concave: farthest extreme; linear: sign of the slope
                            const double dL = fabs(L - xv), dH = fabs(H - xv);
                            const double pickv = (dL >= dH) ? L : H;
                            const bool nearv = fabs(dL - dH) <= 1e-7 * (dL + dH);
                            const double pickl = (t1 > 0.0) ? L : H;
                            const bool nearl = fabs(t1) * (fabs(L) + fabs(H)) <= 1e-9 * fabs(fcur) + 1e-300;
                            pick = (t2 > 0.0) ? pickc : ((t2 < 0.0) ? pickv : pickl);
                            nr = (t2 > 0.0) ? near : ((t2 < 0.0) ? nearv : nearl);
                        }
                        const bool act = !conv;
                        redo = redo || (act && nC > 0 && (slowset != 0 || nr || !(pick == pick)));
                        const double dlt = pick - xi;
                        const bool moved = act && nC > 0 && fabs(dlt) > a.tol;
                        const double delta = moved ? dlt : 0.0;
                        xb[c] = moved ? pick : xi;
                        fcur += delta * (t2 * (pick + xi) + t1);   // f(pick) - f(xi), exact algebra
                        vis += act ? 1 : 0;
                        accn += moved ? 1 : 0;
                        upd = moved ? 0 : upd + 1;           // only read while the restart is live
                        conv = conv || (upd == nlim);
#pragma unroll
                        for (int c2 = c + 1; c2 < 16; c2++) gb[c2] = __builtin_fma(readlane_d(Dreg[c], c2), delta, gb[c2]);
                    }
                    if (__builtin_amdgcn_ballot_w64(redo) == 0ull) {
                        S.fcur = fcur; S.conv = conv; S.upd_counter = upd;
                        S.visits += vis; S.accepted += accn;
#pragma unroll
                        for (int c = 0; c < 16; c++) Xs[(16 * b + c) * 16 + r] = xb[c];
                    } else {
                        // ---- generic loop (rare): the reference's arithmetic, state in LDS
                        pc[6]++;
                        (void)act0;
                        for (int c = 0; c < cmax; c++) {
                            const int64_t i = 16 * b + c;
                            const int slot = c * 16 + r;
                            FeasSet<MAXC> C;
                            C.n = TB.n[slot];
#pragma unroll
                            for (int j = 0; j <= MAXC; j++) { C.lo[j] = TB.lo[j * 256 + slot]; C.hi[j] = TB.hi[j * 256 + slot]; }
                            const double t2 = Dblk[c * 16 + c];
                            const double xi = Xs[i * 16 + r];
                            const double t1 = 2.0 * (G[c * 16 + r] - t2 * xi) + q0b[c];
                            const double t0 = S.fcur - xi * (t2 * xi + t1);
                            DrawKey dk{a.seed, a.first_index + (uint64_t)gr, (uint32_t)i,
                                       (uint32_t)t | 0x80000000u, 0u};
                            double xn = xi;
                            int got = S.conv ? 0 : onevar_minimise<MAXC>(t2, t1, t0, C, dk, &xn);
                            bool moved;
                            double delta;
                            chain_commit<MAXC>(S, got, xn, xi, t2, t1, t0, a.tol, P.n, moved, delta);
                            if (moved) {
                                Xs[i * 16 + r] = xn;
                                for (int c2 = c + 1; c2 < 16; c2++) G[c2 * 16 + r] += Dblk[c * 16 + c2] * delta;
                            }
                        }
                    }
                }
                unsigned long long live = __builtin_amdgcn_ballot_w64(lane < 16 && !S.conv);
                if (lane == 0) *done = (live == 0ull) ? 1 : 0;
            }
            PROF_TICK(3)
            __syncthreads();
            PROF_TICK(4)
            pc[5]++;
            if (*done) { all_done = true; break; }
        }
    }
    __syncthreads();
    if (XLDS)
        for (int64_t idx = tid; idx < n16 * 16; idx += 256) Xg[idx] = Xl[idx];
    if (wave == 0 && lane < 16 && gr < a.R) {
        a.visits[gr] = S.visits; a.accepted[gr] = S.accepted; a.sweeps[gr] = S.sweeps;
        a.status[gr] = S.status;
    }
    if (a.prof && tid == 0)
        for (int k = 0; k < 8; k++) a.prof[tile * 16 + k] = pc[k];
#undef PROF_TICK
}

}  // namespace qcqpmi
