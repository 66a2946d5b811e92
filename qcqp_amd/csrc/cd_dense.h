// Dense-constraint path: problems whose constraints couple coordinates (beamforming, dense
// indefinite QCQPs) at sizes where the one-workgroup-per-tile kernel of cd_general.h is hopeless.
//
//   QuadraticFunction.eval / violations   utilities.py:49-62,133-134   -> dense_products_kernel<1> + dense_viol_kernel
//   get_onevar_func                       utilities.py:99-105          -> dense_products_kernel<0> (+ lazy block correction)
//   get_feasible_intervals, onevar_qcqp   utilities.py:198-288         -> dense_chain_kernel (bounds + gap sweep)
//   coord_descent_phase1 / phase2         qcqp.py:101-178              -> sweep_begin / chain / sweep_end kernels + host loop
//
// Data layout.  ALL m+1 matrices (objective = function 0) live in one array in MFMA A-fragment order,
// block-major:   Gpack[b][k][kk][l] = P_k[16 b + (l & 15)][4 kk + (l >> 4)],
// so that everything a block of 16 coordinates needs -- the 16 rows of every P_k -- is ONE contiguous
// chunk of (m+1) 16 n16 doubles, streamed once per block visit and shared by all tiles of restarts
// (L2 / MALL), and each A operand is a coalesced 512-byte wave load.  The population keeps its
// tile-major layout (kernels.h), which is exactly the B-fragment order.  Everything the chain reads
// per function (G, diagonal blocks, q, tracked f_k) is laid out FUNCTION-FASTEST (padded to m1p, a
// multiple of 64) because the chain's lanes run over functions.
//
// A coordinate sweep is a loop over blocks; per block three launches:
//   products  G[tile][c][r][k] = (P_k[I_b, :] X[tile])[c][r]   16(m+1) x n16 by n16 x 16 per tile on
//             the matrix cores; grid = function groups x tile groups, so the whole chip works on one
//             block even when there are few restarts (cfg5: 32 tiles per GPU but 1025 functions);
//   diag      the 16 x 16 diagonal blocks P_k[I_b, I_b] -> Dg[c][c'][k] (shared by all restarts);
//   chain     ONE WAVEFRONT PER RESTART walks the 16 coordinates in order, lane = function slot
//             (k = lane, lane + 64, ...).  Everything that belongs to the restart (x_i, the moves made
//             so far, slack, bounds) is wave-uniform, so there is no workgroup barrier anywhere: cross
//             lane traffic is DPP reductions, ballots and the wave's own in-order LDS queue.
//             Coordinate c: every lane forms (t2, t1, t0) of its functions -- G corrected by the moves
//             already made inside the block (Gauss-Seidel, through Dg) and t0 from the tracked f_k(x)
//             -- then the feasible set is built WITHOUT sorting all end points:
//               each constraint allows  [lo, hi]  minus at most one gap  (hi0, lo1);
//               L = max lo, H = min hi are wave reductions over the constraints;
//               only gaps that cut into [L, H] survive (typically none or a few), they are sorted
//               and swept by lane 0.
//             The reference's end-point rules (zero-width segments, segments
//             ending at +inf and segments whose right end is shared by two intervals vanish) become
//             a multiplicity count of H and of coincident gap starts.
// Arithmetic differs from the reference by summation order (MFMA) and by tracking f_k incrementally;
// bit-exact parity is the job of cd_general.h's exact mode (n <= 64), this path is checked within
// tolerance / statistically (tests/test_gpu_parity.py).
#pragma once
#include "cd_general.h"

namespace qcqpmi {

constexpr int DN_GC = 64;    // gaps kept per restart
constexpr int DN_SC = 32;    // segments kept per restart
constexpr int DN_WPB = 4;    // at most this many restarts (= waves) per workgroup of the chain kernel
// LDS doubles per wave of the chain kernel besides the per-function arrays: t2, t1, t0, f_k (DN_FARR_MIN) and, when G is
// staged in LDS (few restarts: there is room), the gap of every two-interval function kept from pass 1 (DN_FARR)
constexpr int DN_LDS_WAVE = 2 * DN_GC + 2 * DN_SC + 32 + 8;
constexpr int DN_FARR = 6;
constexpr int DN_FARR_MIN = 4;
constexpr int DN_PF = 8;     // function slots per lane whose per-coordinate operands are requested in one batch

typedef double dn_v4d __attribute__((ext_vector_type(4)));

struct DenseProblem {
    const double *Gpack;   // [NB][m1][KS][64]
    const double *q;       // [m1][n16]
    const double *qT;      // [n16][m1p]
    const double *r;       // [m1]
    const int *relop;      // [m1]  (0 for the objective)
    int64_t n, n16;
    int NB, KS, m1, m1p;
};

// ------------------------------------------------------------------------------------ packing
// src: row-major matrix of function k (objective: P0 with ld n16; constraint: gP + (k-1) n^2, ld n)
__global__ void dense_pack_kernel(const double *__restrict__ P0, const double *__restrict__ gP,
                                  double *__restrict__ Gpack, int64_t n, int64_t n16, int m1, int kbase) {
    const int k = kbase + blockIdx.y;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n16 * n16) return;
    const int KS = (int)(n16 / 4);
    const int l = (int)(idx & 63);
    const int64_t fk = idx >> 6;          // b * KS + kk
    const int64_t b = fk / KS, kk = fk % KS;
    const int64_t row = 16 * b + (l & 15), col = 4 * kk + (l >> 4);
    double v = 0.0;
    if (row < n && col < n) v = (k == 0) ? P0[row * n16 + col] : gP[((int64_t)(k - 1) * n + row) * n + col];
    Gpack[((b * m1 + k) * KS + kk) * 64 + l] = v;
}

// synthetic dense function (qcqpmi_set_quad_generated): element (row, col) of function k
__device__ inline double dense_gen_value(uint64_t seed, int k, int64_t n, int64_t row, int64_t col, double scale,
                                         double diag_add) {
    const int64_t lo = row < col ? row : col, hi = row < col ? col : row;
    const double g = keyed_normal(seed, (1ull << 48) + (uint64_t)k, (uint64_t)(lo * n + hi));
    return (row == col) ? scale * g + diag_add : scale * 0.70710678118654752440 * g;
}

// straight into the block-major fragment layout
__global__ void dense_gen_pack_kernel(double *__restrict__ Gpack, int64_t n, int64_t n16, int m1, int k, uint64_t seed,
                                      double scale, double diag_add) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n16 * n16) return;
    const int KS = (int)(n16 / 4);
    const int l = (int)(idx & 63);
    const int64_t fk = idx >> 6;
    const int64_t b = fk / KS, kk = fk % KS;
    const int64_t row = 16 * b + (l & 15), col = 4 * kk + (l >> 4);
    double v = 0.0;
    if (row < n && col < n) v = dense_gen_value(seed, k, n, row, col, scale, diag_add);
    Gpack[((b * m1 + k) * KS + kk) * 64 + l] = v;
}

// row-major padded (the objective's P0) and the linear term
__global__ void dense_gen_rowmajor_kernel(double *__restrict__ P, int64_t n, int64_t n16, int k, uint64_t seed,
                                          double scale, double diag_add) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n16 * n16) return;
    const int64_t row = idx / n16, col = idx % n16;
    P[idx] = (row < n && col < n) ? dense_gen_value(seed, k, n, row, col, scale, diag_add) : 0.0;
}

__global__ void dense_gen_q_kernel(double *__restrict__ q, int64_t n, int k, uint64_t seed, double qscale) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) q[j] = (qscale == 0.0) ? 0.0 : qscale * keyed_normal(seed, (2ull << 48) + (uint64_t)k, (uint64_t)j);
}

// ----------------------------------------------------------------------------------- products
struct DenseProdArgs {
    DenseProblem D;
    const double *X;        // tile-major population
    int ntiles;
    int b;                  // MODE 0: the block
    int zs;                 // grid.z: MODE 0 splits the contraction (K), MODE 1 splits the row blocks
    double *G;              // MODE 0: [zs][ntiles][16 c][16 r][m1p] partial products (summed by the reader in z order)
    double *F;              // MODE 1: [zs][m1][Rpad] partial quadratic forms
    int64_t Rpad;
    const uint8_t *tile_on; // MODE 0: optional per-tile "any restart sweeping" flags, or nullptr
    int hole;               // MODE 0: stage (block of 16 columns) left out of the contraction, or -1
    int ch_only;            // MODE 0: >= 0: contract over this stage alone and store into plane `zplane` (fix-up)
    int zplane;
};

constexpr int DP_FG = 8;    // functions per workgroup
constexpr int DP_TG = 8;    // tiles of 16 candidates per workgroup
constexpr int DP_KC = 4;    // k-steps per LDS stage (= one block of 16 coordinates)
constexpr int DP_LDS_BYTES = 2 * (DP_FG + DP_TG) * DP_KC * 64 * 8;

typedef double dn_v2d __attribute__((ext_vector_type(2)));

// LDS-tiled fp64 GEMM on v_mfma_f64_16x16x4_f64:  (16 DP_FG rows of the packed matrices) x (16 DP_TG
// candidates) per workgroup, 4 waves as 2 x 2, each wave a register block of 4 functions x 4 tiles
// (16 accumulators).  Both operands are already stored in fragment order, so a stage is 16 contiguous
// 2-KB streams (8 functions, 8 tiles) copied with 16-byte loads; the next stage's loads are in flight
// while the current one multiplies (register prefetch + LDS double buffer, one barrier per stage).
// Per stage and wave: 32 ds_read_b64 feed 64 MFMAs.
//   MODE 0: G of block a.b (contraction split over grid.z);
//   MODE 1: quadratic forms f_k(x) = x' P_k x + q_k' x + r_k (row blocks split over grid.z)
//   MODE 2: ONE matrix (D.m1 == 1, Gpack = the objective's Apack): the 8 streams are 8 ROW BLOCKS of it,
//           x' P x accumulated per row-block group into partial planes F[2 blockIdx.x + wm][Rpad]
//   MODE 3: ONE matrix as in MODE 2, affine map  OUT = mu 1' + M X  stored tile-major (SDR sampling
//           x = mu + F xi, qcqp.py:396): a.G = OUT, D.q = mu
template <int MODE>
__global__ __launch_bounds__(256, 2) void dense_products_kernel(DenseProdArgs a) {
    extern __shared__ double smem[];
    const DenseProblem &D = a.D;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f0 = DP_FG * blockIdx.x, tg0 = DP_TG * blockIdx.y, z = blockIdx.z;
    // narrow tile group (at most 4 tiles of candidates: few restarts, the 47-column factor of the relaxation solver): the
    // four waves split the 8 FUNCTIONS (2 each) and all walk the same 4 tile slots -- with the 2 x 2 arrangement the two
    // waves of the upper tile slots would have nothing to multiply and half of the SIMDs would idle.  (MODE 2 keeps the
    // 2 x 2 arrangement: its partial planes are indexed by the wave row.)
    const bool narrow = MODE != 2 && a.ntiles - tg0 <= 4;
    const int nu = narrow ? 2 : 4;                                   // functions per wave
    const int wm = narrow ? 0 : (wave >> 1), wn = narrow ? 0 : (wave & 1);
    const int fw = narrow ? 2 * wave : 4 * wm;                       // first function of this wave inside the group
    if (MODE == 0 && a.tile_on) {   // uniform for the workgroup
        bool any = false;
        for (int t = 0; t < DP_TG; t++) any = any || (tg0 + t < a.ntiles && a.tile_on[tg0 + t]);
        if (!any) return;
    }
    // this thread's slice of the 16 streams: passes 0..3 -> functions, 4..7 -> tiles
    const int half = tid >> 7, off = (tid & 127) * 2;
    int fs[4], ts[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int f = f0 + 2 * p + half, t = tg0 + 2 * p + half;
        fs[p] = (MODE >= 2) ? (f < D.NB ? f : D.NB - 1) : (f < D.m1 ? f : D.m1 - 1);   // clamped: loaded, multiplied, never stored
        ts[p] = t < a.ntiles ? t : a.ntiles - 1;
    }
    const int kf0 = f0 + fw, tl0 = tg0 + 4 * wn;   // this wave's functions / tiles
    const int vt = (a.ntiles - tl0) < 4 ? ((a.ntiles - tl0) > 0 ? a.ntiles - tl0 : 0) : 4;   // valid tile slots of this wave
    double fa[4][4];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int t = 0; t < 4; t++) fa[u][t] = 0.0;
    const int NBc = D.NB;   // stages per full contraction
    const int b_lo = (MODE == 0) ? a.b : (MODE >= 2 ? 0 : (int)((int64_t)z * D.NB / a.zs));
    const int b_hi = (MODE == 0) ? a.b + 1 : (MODE >= 2 ? 1 : (int)((int64_t)(z + 1) * D.NB / a.zs));
    // MODE 0 walks a LIST of stages: all of them, all but the hole, or the fix-up stage alone
    const int hole = (MODE == 0) ? a.hole : -1, only = (MODE == 0) ? a.ch_only : -1;
    const int nch = only >= 0 ? 1 : (hole >= 0 ? NBc - 1 : NBc);
    const int zsk = only >= 0 ? 1 : a.zs;
    const int ch_lo = (MODE == 0) ? (int)((int64_t)z * nch / zsk) : 0;
    const int ch_hi = (MODE == 0) ? (int)((int64_t)(z + 1) * nch / zsk) : NBc;
    auto stage_of = [&](int j) {   // list position -> stage, clamped to a valid one
        int st = only >= 0 ? only : ((hole >= 0 && j >= hole) ? j + 1 : j);
        return st < NBc ? st : NBc - 1;
    };
    for (int b = b_lo; b < b_hi; b++) {
        dn_v4d acc[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int t = 0; t < 4; t++) acc[u][t] = dn_v4d{0.0, 0.0, 0.0, 0.0};
        const double *src[8];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            src[p] = (MODE >= 2) ? D.Gpack + ((int64_t)fs[p] * D.KS) * 64 + off
                                 : D.Gpack + (((int64_t)b * D.m1 + fs[p]) * D.KS) * 64 + off;
            src[4 + p] = a.X + (int64_t)ts[p] * D.n16 * 16 + off;
        }
        dn_v2d pf[8];
#pragma unroll
        for (int p = 0; p < 8; p++) { pf[p] = dn_v2d{0.0, 0.0}; if (!(narrow && p >= 6)) pf[p] = *reinterpret_cast<const dn_v2d *>(src[p] + (int64_t)stage_of(ch_lo) * 256); }   // narrow: tile slots 4-7 are never multiplied
        int buf = 0;
        __syncthreads();   // the previous row block's readers are done with both buffers
#pragma unroll
        for (int p = 0; p < 8; p++)
            if (!(narrow && p >= 6)) *reinterpret_cast<dn_v2d *>(smem + ((buf * 16 + 2 * p + half) * 256 + off)) = pf[p];
        {
            const int c1 = (ch_lo + 1 < ch_hi) ? ch_lo + 1 : ch_lo;
#pragma unroll
            for (int p = 0; p < 8; p++) if (!(narrow && p >= 6)) pf[p] = *reinterpret_cast<const dn_v2d *>(src[p] + (int64_t)stage_of(c1) * 256);
        }
        __syncthreads();
        // Stage ch multiplies out of `buf`.  The registers hold stage ch+1, fetched a whole stage ago:
        // they go to the other buffer first (no wait), then the fetch of stage ch+2 is issued and
        // stays in flight behind the 64 MFMAs.  All loads are unconditional (clamped).
        for (int ch = ch_lo; ch < ch_hi; ch++) {
#pragma unroll
            for (int p = 0; p < 8; p++)
                if (!(narrow && p >= 6)) *reinterpret_cast<dn_v2d *>(smem + (((buf ^ 1) * 16 + 2 * p + half) * 256 + off)) = pf[p];
            const int chn = (ch + 2 < ch_hi) ? ch + 2 : ch_hi - 1;
#pragma unroll
            for (int p = 0; p < 8; p++) if (!(narrow && p >= 6)) pf[p] = *reinterpret_cast<const dn_v2d *>(src[p] + (int64_t)stage_of(chn) * 256);
            const double *As = smem + (buf * 16 + fw) * 256 + lane;
            const double *Bs = smem + (buf * 16 + 8 + 4 * wn) * 256 + lane;
            if (MODE == 0) {
                double av[DP_KC][4], bv[DP_KC][4];
#pragma unroll
                for (int ks = 0; ks < DP_KC; ks++)
#pragma unroll
                    for (int u = 0; u < 4; u++) { av[ks][u] = As[u * 256 + ks * 64]; bv[ks][u] = Bs[u * 256 + ks * 64]; }
#pragma unroll
                for (int ks = 0; ks < DP_KC; ks++)
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        if (t >= vt) continue;   // wave-uniform: tile slot beyond the population (small populations)
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            if (u >= nu) continue;   // wave-uniform (narrow tile group)
                            acc[u][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks][u], bv[ks][t], acc[u][t], 0, 0, 0);
                        }
                    }
            } else {
                // the quadratic-form accumulators need 32 more registers: operands two k-steps at a time
#pragma unroll
                for (int k2 = 0; k2 < DP_KC; k2 += 2) {
                    double av[2][4], bv[2][4];
#pragma unroll
                    for (int ks = 0; ks < 2; ks++)
#pragma unroll
                        for (int u = 0; u < 4; u++) { av[ks][u] = As[u * 256 + (k2 + ks) * 64]; bv[ks][u] = Bs[u * 256 + (k2 + ks) * 64]; }
#pragma unroll
                    for (int ks = 0; ks < 2; ks++)
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            if (t >= vt) continue;   // wave-uniform
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                if (u >= nu) continue;   // wave-uniform (narrow tile group)
                                acc[u][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks][u], bv[ks][t], acc[u][t], 0, 0, 0);
                            }
                        }
                }
            }
            __syncthreads();
            buf ^= 1;
        }
        if (MODE == 0) {
            // D layout: register v of lane l = row c = (l >> 4) + 4 v, column r = l & 15
            double *Gz = a.G + (int64_t)(only >= 0 ? a.zplane : z) * a.ntiles * 256 * D.m1p;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                if (tl0 + t >= a.ntiles) continue;
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int c = (lane >> 4) + 4 * v, r = lane & 15;
                    double *g = Gz + (((int64_t)(tl0 + t) * 16 + c) * 16 + r) * D.m1p + kf0;
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        if (u < nu && kf0 + u < D.m1) g[u] = acc[u][t][v];
                }
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (u >= nu || kf0 + u >= D.NB) continue;   // wave-uniform
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    if (tl0 + t >= a.ntiles) continue;
                    double *xo = a.G + (int64_t)(tl0 + t) * D.n16 * 16;
#pragma unroll
                    for (int v = 0; v < 4; v++) {
                        const int64_t i = 16 * (int64_t)(kf0 + u) + (lane >> 4) + 4 * v;
                        xo[i * 16 + (lane & 15)] = (i < D.n) ? D.q[i] + acc[u][t][v] : 0.0;
                    }
                }
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (u >= nu || kf0 + u >= D.NB) continue;   // wave-uniform
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int64_t i = 16 * (int64_t)(kf0 + u) + (lane >> 4) + 4 * v;
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const int tt = (tl0 + t < a.ntiles) ? tl0 + t : a.ntiles - 1;
                        fa[0][t] += a.X[(int64_t)tt * D.n16 * 16 + i * 16 + (lane & 15)] * acc[u][t][v];
                    }
                }
            }
        } else {
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int64_t i = 16 * (int64_t)b + (lane >> 4) + 4 * v;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int tt = (tl0 + t < a.ntiles) ? tl0 + t : a.ntiles - 1;
                    const double xi = a.X[(int64_t)tt * D.n16 * 16 + i * 16 + (lane & 15)];
#pragma unroll
                    for (int u = 0; u < 4; u++) fa[u][t] += xi * acc[u][t][v];   // q_k' x: dense_linear_kernel
                }
            }
        }
    }
    if (MODE == 2) {
        double *Fz = a.F + (int64_t)(2 * blockIdx.x + wm) * a.Rpad;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            double s = fa[0][t];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (lane < 16 && tl0 + t < a.ntiles) Fz[(int64_t)(tl0 + t) * 16 + lane] = s;
        }
    }
    if (MODE == 1) {
        double *Fz = a.F + (int64_t)z * D.m1 * a.Rpad;
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                double s = fa[u][t];
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                if (lane < 16 && u < nu && kf0 + u < D.m1 && tl0 + t < a.ntiles)
                    Fz[(int64_t)(kf0 + u) * a.Rpad + (int64_t)(tl0 + t) * 16 + lane] = s + (z == 0 ? D.r[kf0 + u] : 0.0);
            }
    }
}

// S = sum_k w_k P_k in the fragment layout of ONE matrix ([b][kk][lane], the objective's Apack layout): one
// streaming pass over all matrices (HBM-bound: (m+1) n16^2 8 bytes), used by the general SDP solver, whose
// gradient is 2 S V
__global__ __launch_bounds__(256) void dense_wsum_pack_kernel(DenseProblem D, const double *__restrict__ w,
                                                              double *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (b * KS + kk) * 64 + lane
    if (idx >= D.n16 * D.n16) return;
    const int64_t per_b = (int64_t)D.KS * 64;
    const int64_t b = idx / per_b, rem = idx % per_b;
    const double *src = D.Gpack + (b * D.m1) * per_b + rem;
    double s0 = 0.0, s1 = 0.0;
    int k = 0;
    // Zero weights are skipped (wave-uniform branches on scalar loads): adding 0 P_k changes nothing, and in the relaxation
    // solver -- w = the multipliers of an augmented Lagrangian -- all but a handful of the m + 1 weights are zero for most of
    // the run (median 2 of 513 at n = 2048): the pass over ALL matrices (137.6 GB, 22 ms at full size) becomes a pass over
    // the active ones.  The two accumulators keep their association (even / odd function index).
    for (; k + 1 < D.m1; k += 2) {
        const double w0 = w[k], w1 = w[k + 1];
        if (w0 != 0.0) s0 = __builtin_fma(w0, src[(int64_t)k * per_b], s0);
        if (w1 != 0.0) s1 = __builtin_fma(w1, src[(int64_t)(k + 1) * per_b], s1);
    }
    if (k < D.m1 && w[k] != 0.0) s0 = __builtin_fma(w[k], src[(int64_t)k * per_b], s0);
    out[idx] = s0 + s1;
}

// one matrix in fragment layout [b][kk][lane] -> row-major n x n
__global__ void dense_unpack_kernel(const double *__restrict__ Spack, double *__restrict__ out, int64_t n, int KS) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    const int64_t row = idx / n, col = idx % n;
    const int64_t b = row >> 4, kk = col >> 2;
    const int l = (int)(((col & 3) << 4) | (row & 15));
    out[idx] = Spack[(b * KS + kk) * 64 + l];
}

// linear terms  lin[k][gr] = q_k' x_gr  (2 m1 n flops per candidate: noise next to the quadratic forms)
__global__ __launch_bounds__(256) void dense_linear_kernel(DenseProblem D, const double *__restrict__ X, int64_t Rpad,
                                                           double *__restrict__ lin) {
    const int r = threadIdx.x & 15, k = blockIdx.y * 16 + (threadIdx.x >> 4);
    const int64_t tile = blockIdx.x;
    if (k >= D.m1) return;
    const double *q = D.q + (int64_t)k * D.n16, *x = X + tile * D.n16 * 16 + r;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int64_t j = 0; j < D.n16; j += 4) {
        s0 = __builtin_fma(q[j], x[j * 16], s0);
        s1 = __builtin_fma(q[j + 1], x[(j + 1) * 16], s1);
        s2 = __builtin_fma(q[j + 2], x[(j + 2) * 16], s2);
        s3 = __builtin_fma(q[j + 3], x[(j + 3) * 16], s3);
    }
    lin[(int64_t)k * Rpad + tile * 16 + r] = (s0 + s1) + (s2 + s3);
}

// partial planes of the quadratic forms -> plane 0 (fixed order), without the linear terms
__global__ void dense_sum_planes_kernel(double *__restrict__ F, int zs, int m1, int64_t Rpad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)m1 * Rpad) return;
    double f = F[idx];
    for (int z = 1; z < zs; z++) f += F[(int64_t)z * m1 * Rpad + idx];
    F[idx] = f;
}

// f0 and the maximum violation of every candidate from the table of function values; also the
// restart-major copy Ft[gr][k] the chain kernel tracks
__global__ void dense_viol_kernel(double *__restrict__ F, int zs, const double *__restrict__ lin, const int *__restrict__ relop, int m1, int m1p,
                                  int64_t Rpad, double *__restrict__ f0, double *__restrict__ maxviol,
                                  double *__restrict__ Ft) {
    const int64_t gr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gr >= Rpad) return;
    double v = -QM_INF;
    for (int k = 0; k < m1; k++) {
        double f = F[(int64_t)k * Rpad + gr];
        for (int z = 1; z < zs; z++) f += F[((int64_t)z * m1 + k) * Rpad + gr];   // partial planes, fixed order
        f += lin[(int64_t)k * Rpad + gr];
        F[(int64_t)k * Rpad + gr] = f;
        if (Ft) Ft[gr * m1p + k] = f;
        if (k == 0) continue;
        const double w = (relop[k] == RELOP_EQ) ? fabs(f) : (f > 0.0 ? f : 0.0);
        v = w > v ? w : v;
    }
    f0[gr] = F[gr];
    maxviol[gr] = v;
}

// objective and maximum violation of the final points from the function values the chain has tracked through every
// accepted move (f_k += delta (t2 (xn + xi) + t1)): saves the evaluation pass over all matrices after phase 2 -- as
// expensive as the products of one whole sweep.  Agrees with a fresh evaluation to ~1e-12 relative (tested at 1e-9).
__global__ void dense_tracked_out_kernel(const double *__restrict__ Ft, const int *__restrict__ relop, int m1, int m1p, int64_t R,
                                         double *__restrict__ f0, double *__restrict__ maxviol) {
    const int64_t gr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gr >= R) return;
    double v = -QM_INF;
    for (int k = 1; k < m1; k++) {
        const double f = Ft[gr * m1p + k];
        const double w = (relop[k] == RELOP_EQ) ? fabs(f) : (f > 0.0 ? f : 0.0);
        v = w > v ? w : v;
    }
    f0[gr] = Ft[gr * m1p];
    maxviol[gr] = v;
}

// diagonal blocks of every function for block b:  Dg[c][c'][k] = P_k[16 b + c][16 b + c']
__global__ __launch_bounds__(256) void dense_diag_kernel(DenseProblem D, int b, double *__restrict__ Dg) {
    const int k = blockIdx.x, t = threadIdx.x;
    const int c = t >> 4, c2 = t & 15;
    const int kk = 4 * b + (c2 >> 2), l = (c2 & 3) * 16 + c;
    Dg[(int64_t)t * D.m1p + k] = D.Gpack[(((int64_t)b * D.m1 + k) * D.KS + kk) * 64 + l];
}

// -------------------------------------------------------------------------------- sweep state
struct DenseState {          // one entry per restart (Rpad), persistent across the block launches
    int64_t *upd, *visits, *accepted, *sweeps;
    uint8_t *live, *on;      // live: still iterating; on: takes part in the current sweep
    double *viol_last;
    int *status;
    uint8_t *tile_on;        // [ntiles] any restart of the tile is on
    int *nlive;              // [1]
};

template <int PHASE>
__global__ void dense_sweep_begin_kernel(DenseState S, int64_t R, int64_t Rpad, double viol_tol) {
    const int64_t gr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // blockDim = 64 = 4 tiles
    bool live = false;
    if (gr < R) {
        live = S.live[gr] != 0;
        if (PHASE == 1 && live && S.viol_last[gr] < viol_tol) live = false;   // qcqp.py:111
        S.live[gr] = live ? 1 : 0;
        S.on[gr] = live ? 1 : 0;
        if (live) S.sweeps[gr]++;
    }
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(live);
    const int lane = threadIdx.x & 63;
    if ((lane & 15) == 0 && gr < Rpad) S.tile_on[gr >> 4] = ((bal >> lane) & 0xffffull) ? 1 : 0;
    if (lane == 0 && bal) atomicAdd(S.nlive, __builtin_popcountll(bal));
}

__global__ void dense_state_init_kernel(DenseState S, const uint8_t *flag, int64_t R, int64_t Rpad) {
    const int64_t gr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gr >= Rpad) return;
    S.upd[gr] = 0; S.visits[gr] = 0; S.accepted[gr] = 0; S.sweeps[gr] = 0; S.status[gr] = 0;
    S.viol_last[gr] = QM_INF;
    S.live[gr] = (gr < R && (!flag || flag[gr])) ? 1 : 0;
    S.on[gr] = 0;
}

// phase 1, end of a sweep: viol = max(prob.violations(x)) (qcqp.py:142) from the tracked f_k
__global__ void dense_sweep_end_kernel(DenseState S, const double *__restrict__ Ft, const int *__restrict__ relop,
                                       int m1, int m1p, int64_t R) {
    const int64_t gr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gr >= R || !S.live[gr]) return;
    double v = -QM_INF;
    for (int k = 1; k < m1; k++) {
        const double f = Ft[gr * m1p + k];
        const double w = (relop[k] == RELOP_EQ) ? fabs(f) : (f > 0.0 ? f : 0.0);
        v = w > v ? w : v;
    }
    S.viol_last[gr] = v;
}

// ------------------------------------------------------------------------- wave-level helpers
// cross-lane traffic through the wave's LDS region: the hardware keeps a wave's LDS operations in
// order, the fence keeps the compiler from reordering them
__device__ inline void dn_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int CTRL, int ROW_MASK>
__device__ inline double dn_dpp(double v) {
    // lanes without a source lane (or masked rows) keep their own value: harmless for max / min
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ inline double dn_bcast63(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

__device__ inline double dn_bcast0(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

#define DN_WAVE_REDUCE(OP)                                             \
    { double w_;                                                       \
      w_ = dn_dpp<0x111, 0xf>(v); v = OP(v, w_);   /* row_shr:1 */     \
      w_ = dn_dpp<0x112, 0xf>(v); v = OP(v, w_);   /* row_shr:2 */     \
      w_ = dn_dpp<0x114, 0xf>(v); v = OP(v, w_);   /* row_shr:4 */     \
      w_ = dn_dpp<0x118, 0xf>(v); v = OP(v, w_);   /* row_shr:8 */     \
      w_ = dn_dpp<0x142, 0xa>(v); v = OP(v, w_);   /* row_bcast:15 */  \
      w_ = dn_dpp<0x143, 0xc>(v); v = OP(v, w_);   /* row_bcast:31 */  \
      return dn_bcast63(v); }

__device__ inline double dn_max2(double a, double b) { return a > b ? a : b; }
__device__ inline double dn_min2(double a, double b) { return a < b ? a : b; }
__device__ inline double dn_wave_max(double v) DN_WAVE_REDUCE(dn_max2)
__device__ inline double dn_wave_min(double v) DN_WAVE_REDUCE(dn_min2)

// --------------------------------------------------------------------------------- chain kernel
struct DenseChainArgs {
    DenseProblem D;
    double *X;
    int64_t R, Rpad;
    const double *G;       // [zs][ntiles][16][16][m1p] partial products
    int zs;
    int64_t gz_stride;
    const double *Dg;      // [16][16][m1p]
    double *Ft;            // [Rpad][m1p] tracked function values
    const double *slack;   // [Rpad] phase 2: the fixed slack
    DenseState S;
    int b;
    int64_t t;             // sweep number
    double tol, viol_tol;
    uint64_t seed, first_index;
    long long *prof;       // optional: 16 tick sums of the wave of restart 0 (qcqpmi_debug_dense_profile), or nullptr
    int mw_Tc, mw_ts;      // dense_chain_mw_kernel: threads that hold constraints, index of the serial thread (mw_geometry)
    int c_lo = 0, c_hi = 16;   // dense_chain_mw_kernel: coordinates of the block to visit (the unit step of the parity tests visits one)
};

// stage timer of one wave (s_memtime ticks): 0 set-up, 1 coefficients, 2 bounds + reductions, 3 gaps, 4 segment sweep
// (lane 0), 5 minimiser / draw (lane 0), 6 commit, 7 write-back, 8 coordinates visited, 9 feasible-set evaluations
#ifndef DN_PROFILE
#define DN_PROFILE 0       // build with -DDN_PROFILE=1 to compile the stage timers in (s_memtime drains the wave's loads)
#endif
struct DnProf {
    long long t[10];
    long long last;
    bool on;
    __device__ inline void start(bool enable) {
        on = DN_PROFILE && enable;
        for (int i = 0; i < 10; i++) t[i] = 0;
        last = on ? (long long)__builtin_amdgcn_s_memtime() : 0;
    }
    __device__ inline void tick(int slot) {
        if (DN_PROFILE && on) { const long long now = (long long)__builtin_amdgcn_s_memtime(); t[slot] += now - last; last = now; }
    }
};

struct DnWave {            // the wave's LDS region
    double *t2, *t1, *t0, *F;    // [m1p]
    double *fga, *fgb;           // [m1p] the gap of a function that allows two intervals (pass 1 -> pass 2)
    double *gapa, *gapb;         // [DN_GC]
    double *seglo, *seghi;       // [DN_SC]
    double *xb, *dlt;            // [16]
    int *misc;                   // [0] number of segments
    double *G;                   // GLDS: [16][m1p]
};

// lane 0: [L, H] minus the sorted gaps -> segment list, filtered with the end-point rules
__device__ inline int dn_sweep_segments(double *ga, double *gb, int ng, double *slo, double *shi, double L, double H,
                                        int mH, int *overflow) {
    for (int i = 1; i < ng; i++) {   // insertion sort by gap start
        const double a = ga[i], b = gb[i];
        int j = i - 1;
        while (j >= 0 && ga[j] > a) { ga[j + 1] = ga[j]; gb[j + 1] = gb[j]; j--; }
        ga[j + 1] = a; gb[j + 1] = b;
    }
    int ns = 0, idx = 0;
    double cur = L;
    bool closed = false;   // cur ran past H
    while (idx < ng) {
        const double a = ga[idx];
        if (a >= cur) {
            int cnt = 0, j = idx;
            double nb = cur;
            while (j < ng && ga[j] == a) { cnt++; nb = gb[j] > nb ? gb[j] : nb; j++; }
            if (a == H) cnt += mH;
            if (cur != a && cnt == 1) {
                if (ns < DN_SC) { slo[ns] = cur; shi[ns] = a; ns++; } else *overflow = 1;
            }
            cur = nb;
            idx = j;
        } else {
            cur = gb[idx] > cur ? gb[idx] : cur;
            idx++;
        }
        if (cur > H) { closed = true; break; }
    }
    if (!closed && cur <= H && cur != H && mH == 1) {
        if (ns < DN_SC) { slo[ns] = cur; shi[ns] = H; ns++; } else *overflow = 1;
    }
    return ns;
}

// Feasible set of the current coordinate at slack s from the coefficient arrays in LDS.
// Returns the number of segments (wave-uniform); the list is in W.seglo / W.seghi.
template <bool SG>      // SG: the gaps found in pass 1 are kept in W.fga / W.fgb; else pass 2 recomputes them
__device__ inline int dn_feasible_set(const DnWave &W, const DenseProblem &D, int lane, double s, int *overflow, DnProf &pf,
                                      unsigned long long relbits) {
    // pass 1: bounds of this lane's constraints
    double L = -QM_INF, H = QM_INF;
    int mH = 0;
    bool empty = false;
    unsigned n2 = 0;   // bit j: function lane + 64 j allows two intervals
    int j = 0;
    for (int k = lane; k < D.m1; k += 64, j++) {
        if (k == 0) continue;
        const double t2 = W.t2[k], t1 = W.t1[k];
        if (t2 == 0.0 && t1 == 0.0) continue;   // qcqp.py:116,166
        const Seg2 iv = feasible_intervals(t2, t1, W.t0[k], (int)((relbits >> (2 * j)) & 3ull), s);
        if (iv.n == 0) { empty = true; continue; }
        const double lo = iv.lo0, hi = (iv.n == 2) ? iv.hi1 : iv.hi0;
        if (iv.n == 2) { n2 |= 1u << j; if (SG) { W.fga[k] = iv.hi0; W.fgb[k] = iv.lo1; } }   // same lane reads them back in pass 2
        L = lo > L ? lo : L;
        if (hi < H) { H = hi; mH = 1; } else if (hi == H) mH++;
    }
    const double Lg = dn_wave_max(L), Hg = dn_wave_min(H);
    const bool anyempty = __builtin_amdgcn_ballot_w64(empty) != 0ull;
    // multiplicity of Hg; the base interval (-inf, +inf) is one more interval ending at +inf
    int mHg = (Hg == QM_INF) ? 1 : 0;
    {
        unsigned long long mk = __builtin_amdgcn_ballot_w64(H == Hg && mH > 0);
        while (mk) {
            const int l = __builtin_ctzll(mk);
            mk &= mk - 1;
            mHg += __builtin_amdgcn_readlane(mH, l);
        }
    }
    pf.tick(2);
    // pass 2: gaps that cut into [Lg, Hg]  (two-interval constraints only)
    int ng = 0;
    const int kpt = (D.m1 + 63) >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (__builtin_amdgcn_ballot_w64(n2 != 0u) != 0ull) {
        for (int jj = 0; jj < kpt; jj++) {
            const int k = lane + 64 * jj;
            bool has = false;
            double ga = 0.0, gb = 0.0;
            if ((n2 >> jj) & 1u) {
                if (SG) { ga = W.fga[k]; gb = W.fgb[k]; }
                else {
                    const Seg2 iv = feasible_intervals(W.t2[k], W.t1[k], W.t0[k], (int)((relbits >> (2 * jj)) & 3ull), s);
                    ga = iv.hi0; gb = iv.lo1;
                }
                has = gb > Lg && ga <= Hg;
            }
            const unsigned long long mk = __builtin_amdgcn_ballot_w64(has);
            if (has) {
                const int pos = ng + __builtin_popcountll(mk & lt);
                if (pos < DN_GC) { W.gapa[pos] = ga; W.gapb[pos] = gb; }
            }
            ng += __builtin_popcountll(mk);
        }
    }
    if (ng > DN_GC) { *overflow = 1; ng = DN_GC; }
    dn_wave_sync();
    pf.tick(3);
    if (lane == 0) {
        int ns = 0;
        if (!anyempty && Lg <= Hg) ns = dn_sweep_segments(W.gapa, W.gapb, ng, W.seglo, W.seghi, Lg, Hg, mHg, overflow);
        W.misc[0] = ns;
    }
    dn_wave_sync();
    const int nsu = __builtin_amdgcn_readfirstlane(W.misc[0]);
    if (pf.on) pf.t[9]++;
    pf.tick(4);
    return nsu;
}

// entry (c2, c) of the diagonal blocks: Dg[16 c2 + c][k]
__device__ inline double Dc2(const double *Dg, int c2, int c, int m1p, int k) { return Dg[((int64_t)c2 * 16 + c) * m1p + k]; }

// GLDS: the restart's 16 rows of G (K-split partials summed) are staged in LDS in one burst of
// coalesced loads at kernel start instead of being fetched coordinate by coordinate.
template <int PHASE, bool GLDS>
__global__ __launch_bounds__(64 * DN_WPB) void dense_chain_kernel(DenseChainArgs a) {
    extern __shared__ double smem[];
    const DenseProblem &D = a.D;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpb = (int)(blockDim.x >> 6);
    const int64_t gr = (int64_t)blockIdx.x * wpb + wave;
    if (gr >= a.R) return;            // no workgroup barrier anywhere below: waves are independent
    if (!a.S.on[gr]) return;
    const int m1 = D.m1, m1p = D.m1p;
    DnProf pf;
    pf.start(a.prof != nullptr && gr == 0);
    DnWave W;
    {
        double *sp = smem + (size_t)wave * ((GLDS ? 16 + DN_FARR : DN_FARR_MIN) * (size_t)m1p + DN_LDS_WAVE);
        W.t2 = sp; sp += m1p; W.t1 = sp; sp += m1p; W.t0 = sp; sp += m1p; W.F = sp; sp += m1p;
        W.fga = nullptr; W.fgb = nullptr;
        if (GLDS) { W.fga = sp; sp += m1p; W.fgb = sp; sp += m1p; }
        W.gapa = sp; sp += DN_GC; W.gapb = sp; sp += DN_GC;
        W.seglo = sp; sp += DN_SC; W.seghi = sp; sp += DN_SC;
        W.xb = sp; sp += 16; W.dlt = sp; sp += 16;
        W.misc = (int *)sp; sp += 8;
        W.G = sp;
    }
    const int b = a.b;
    const int64_t tile = gr >> 4;
    const int r = (int)(gr & 15);
    double *Xt = a.X + tile * D.n16 * 16;
    double *Ftr = a.Ft + gr * m1p;
    for (int k = lane; k < m1; k += 64) W.F[k] = Ftr[k];
    // relop of the lane's functions, 2 bits per slot (m1p <= 2048)
    unsigned long long relbits = 0ull;
    {
        int j = 0;
        for (int k = lane; k < m1; k += 64, j++) relbits |= (unsigned long long)(D.relop[k] & 3) << (2 * j);
    }
    if (GLDS) {
        const double *Gr = a.G + (tile * 256 + r) * m1p;   // row c of this restart: + c * 16 * m1p
        const int tot2 = 8 * m1p;                          // pairs of doubles (m1p is a multiple of 64: rows stay 16-byte aligned)
#pragma unroll 4
        for (int idx = lane; idx < tot2; idx += 64) {
            const int c = (2 * idx) / m1p, k = 2 * idx - c * m1p;
            const dn_v2d *g0 = reinterpret_cast<const dn_v2d *>(Gr + (int64_t)c * 16 * m1p + k);
            dn_v2d gz[8];
#pragma unroll
            for (int z = 0; z < 8; z++) gz[z] = (z < a.zs) ? g0[(int64_t)z * (a.gz_stride / 2)] : dn_v2d{0.0, 0.0};
            dn_v2d g = gz[0];
#pragma unroll
            for (int z = 1; z < 8; z++) g += gz[z];   // K-split partials, fixed order (absent planes add 0)
            *reinterpret_cast<dn_v2d *>(W.G + 2 * idx) = g;
        }
    }
    if (lane < 16) { W.xb[lane] = Xt[(16 * (int64_t)b + lane) * 16 + r]; W.dlt[lane] = 0.0; }
    // per-restart state: wave-uniform
    bool live = a.S.live[gr] != 0, on = true;
    int64_t upd = a.S.upd[gr], visits = a.S.visits[gr], accepted = a.S.accepted[gr];
    int status = a.S.status[gr], overflow = 0;
    const double slack = (PHASE == 2) ? a.slack[gr] : 0.0;
    unsigned mvmask = 0;   // coordinates of this block that moved
    dn_wave_sync();
    const int cmax = (D.n - 16 * (int64_t)b) < 16 ? (int)(D.n - 16 * (int64_t)b) : 16;
    const SegList SL{W.seglo, W.seghi, nullptr, 0};
    pf.tick(0);

    for (int c = 0; c < cmax && on; c++) {
        const int64_t i = 16 * (int64_t)b + c;
        const double xi = W.xb[c];
        if (pf.on) pf.t[8]++;
        // ---- A. one-variable coefficients of the lane's functions (utilities.py:99-105)
        const double *Gc = a.G + ((tile * 16 + c) * 16 + r) * m1p;
        const double *Dc = a.Dg + (int64_t)(c * 16) * m1p;
        const double *qi = D.qT + i * m1p;
        double vloc = -QM_INF;
        bool inv = false;
        // the diagonal entry and the linear coefficient of the first DN_PF slots are requested in one batch (one round
        // trip instead of one per slot)
        double pt2[DN_PF], pq[DN_PF];
#pragma unroll
        for (int j = 0; j < DN_PF; j++) {
            const int k = lane + 64 * j;
            const bool in = k < m1;
            pt2[j] = in ? Dc[(int64_t)c * m1p + k] : 0.0;
            pq[j] = in ? qi[k] : 0.0;
        }
        {
            int j = 0;
            for (int k = lane; k < m1; k += 64, j++) {
                double g;
                if (GLDS) g = W.G[c * m1p + k];            // kept current by the commits of this block (below)
                else {
                    g = Gc[k];
                    for (int z = 1; z < a.zs; z++) g += Gc[(int64_t)z * a.gz_stride + k];   // K-split partials, fixed order
                    unsigned mm = mvmask;
                    while (mm) {   // Gauss-Seidel inside the block: moves made so far, in coordinate order
                        const int c2 = __builtin_ctz(mm);
                        mm &= mm - 1;
                        g = __builtin_fma(Dc[(int64_t)c2 * m1p + k], W.dlt[c2], g);
                    }
                }
                double t2, ql;
                if (j < DN_PF) {
                    // static indexing of the register batch (j is wave-uniform)
                    t2 = pt2[0]; ql = pq[0];
#pragma unroll
                    for (int u = 1; u < DN_PF; u++) if (j == u) { t2 = pt2[u]; ql = pq[u]; }
                } else { t2 = Dc[(int64_t)c * m1p + k]; ql = qi[k]; }
                const double t1 = 2.0 * (g - t2 * xi) + ql;
                const double t0 = W.F[k] - xi * (t2 * xi + t1);
                W.t2[k] = t2; W.t1[k] = t1; W.t0[k] = t0;
                if (PHASE == 1 && k > 0 && !(t2 == 0.0 && t1 == 0.0)) {
                    const double f = xi * (t2 * xi + t1) + t0;
                    const double v = ((int)((relbits >> (2 * j)) & 3ull) == RELOP_EQ) ? fabs(f) : (f > 0.0 ? f : 0.0);
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
            // ---- B. feasible set at the fixed slack, minimiser of the scalar objective
            const int ns = dn_feasible_set<GLDS>(W, D, lane, slack, &overflow, pf, relbits);
            int got = 0;
            if (lane == 0) {
                SegList C = SL;
                C.n = ns;
                DrawKey dk{a.seed, a.first_index + (uint64_t)gr, (uint32_t)i, (uint32_t)a.t | 0x80000000u, 0u};
                got = general_minimise(W.t2[0], W.t1[0], W.t0[0], C, dk, &xn);
            }
            got = __builtin_amdgcn_readfirstlane(got);
            xn = dn_bcast0(xn);
            if (got < 0) { status = got; live = false; on = false; }
            else if (got && fabs(xn - xi) > a.tol) { moved = true; upd = 0; accepted++; }
            else {
                upd++;
                if (upd == D.n) { live = false; on = false; }   // converged (qcqp.py:172-176)
            }
        } else {
            // ---- B. smallest achievable slack by bisection (qcqp.py:117-131)
            const double viol = dn_wave_max(vloc);
            if (__builtin_amdgcn_ballot_w64(inv) == 0ull) { status = -3; live = false; on = false; }   // ValueError (qcqp.py:117)
            else {
                double new_viol = viol, ss = -a.tol, es = viol - a.viol_tol;
                uint32_t it = 0;
                // Only the last successful step decides the point and the keyed draws are independent of
                // each other: a successful step leaves its segment list in LDS (failed steps write nothing)
                // and the Philox draw happens once, after the bisection.  A list with an unbounded piece
                // draws at once (the reference may raise there).
                int ns_p = 0;
                uint32_t it_p = 0;
                bool pending = false;
                while (es - ss > a.tol) {
                    const double sm = (ss + es) / 2.0;
                    const int ns = dn_feasible_set<GLDS>(W, D, lane, sm, &overflow, pf, relbits);
                    const uint32_t itc = it++;
                    if (ns == 0) { ss = sm; continue; }
                    bool unb = false;
                    if (lane == 0)
                        for (int j = 0; j < ns; j++) unb = unb || __builtin_isinf(W.seglo[j]) || __builtin_isinf(W.seghi[j]);
                    if (__builtin_amdgcn_readfirstlane((int)unb)) {
                        int got = 0;
                        double xc = 0.0;
                        if (lane == 0) {
                            SegList C = SL;
                            C.n = ns;
                            DrawKey dk{a.seed, a.first_index + (uint64_t)gr, (uint32_t)i, (uint32_t)a.t, itc};
                            got = general_minimise(0.0, 0.0, 0.0, C, dk, &xc);
                        }
                        got = __builtin_amdgcn_readfirstlane(got);
                        xc = dn_bcast0(xc);
                        if (got < 0) { status = got; live = false; on = false; pending = false; break; }
                        xn = xc; pending = false;
                    } else {
                        ns_p = ns; it_p = itc; pending = true;
                    }
                    new_viol = sm; es = sm;
                }
                if (pending) {
                    double xc = 0.0;
                    if (lane == 0) {
                        SegList C = SL;
                        C.n = ns_p;
                        DrawKey dk{a.seed, a.first_index + (uint64_t)gr, (uint32_t)i, (uint32_t)a.t, it_p};
                        (void)general_minimise(0.0, 0.0, 0.0, C, dk, &xc);
                    }
                    xn = dn_bcast0(xc);
                }
                if (status == 0) {
                    if (new_viol < viol) { moved = true; upd = 0; accepted++; }
                    else {
                        upd++;
                        if (upd == D.n) on = false;   // "failed": leaves this sweep only (qcqp.py:138-141)
                    }
                }
            }
        }
        pf.tick(5);
        // ---- C. commit: x_i, the block-local move list, f_k(x) += delta (t2 (xn + xi) + t1)
        if (moved) {
            const double d = xn - xi;
            if (lane == 0) { W.xb[c] = xn; W.dlt[c] = d; }
            mvmask |= 1u << c;
            for (int k = lane; k < m1; k += 64) W.F[k] += d * (W.t2[k] * (xn + xi) + W.t1[k]);
            if (GLDS) {
                // Gauss-Seidel inside the block: the rows of the coordinates still to come take the move at once --
                // the same fused multiply-adds in the same order as adding the moves made so far at every visit, but
                // the loads are independent of each other and go out together
                for (int k = lane; k < m1; k += 64) {
#pragma unroll 5
                    for (int c2 = c + 1; c2 < cmax; c2++)
                        W.G[c2 * m1p + k] = __builtin_fma(Dc2(a.Dg, c2, c, m1p, k), d, W.G[c2 * m1p + k]);
                }
            }
            dn_wave_sync();
        }
        pf.tick(6);
    }
    dn_wave_sync();
    if (lane < 16) Xt[(16 * (int64_t)b + lane) * 16 + r] = W.xb[lane];
    for (int k = lane; k < m1; k += 64) Ftr[k] = W.F[k];
    if (lane == 0) {
        a.S.live[gr] = live ? 1 : 0; a.S.on[gr] = on ? 1 : 0;
        a.S.upd[gr] = upd; a.S.visits[gr] = visits; a.S.accepted[gr] = accepted;
        a.S.status[gr] = overflow ? -4 : status;
    }
    if (pf.on) {
        pf.tick(7);
        if (lane == 0) for (int i = 0; i < 10; i++) a.prof[i + (PHASE == 1 ? 0 : 16)] += pf.t[i];
    }
}

}  // namespace qcqpmi
