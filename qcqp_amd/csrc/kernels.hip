// Hand-written gfx950 kernels of the Suggest-and-Improve engine.  fp64 throughout.
//
//   pack_A            P0 (row-major) -> MFMA A-fragment order
//   to_tiles/from_tiles  host column-per-candidate layout <-> tile-major population
//   randn_tiles       RANDOM suggest, batched (qcqp.py:381-382)
//   affine_tiles      X = mu 1^T + F Xi   (SDR sampling, qcqp.py:396; F = factor of Sigma)
//   eval_kernel       f0(x), max violation for a tile of 16 candidates (utilities.py:49-62,133-134)
//   cd_phase1_sep     coordinate descent phase 1, separable constraints (qcqp.py:101-148)
//   cd_phase2_kernel  coordinate descent phase 2 as blocked Gauss-Seidel on v_mfma_f64_16x16x4_f64
//                     (qcqp.py:152-178)
//   select_best       lexicographic (violation bucket, objective) argmin (utilities.py:135-146)
//
// Work decomposition: one workgroup owns one TILE of 16 candidates (see kernels.h); the 16x16
// fp64 MFMA tile is (16 coordinates of a block) x (16 candidates).
#include "kernels.h"
#include "onevar.h"
#include "cd_phase1_sep.h"

namespace qcqpmi {

typedef double v4d __attribute__((ext_vector_type(4)));

// ----------------------------------------------------------------------------- layout kernels

__global__ void pack_A_kernel(const double *__restrict__ P, double *__restrict__ Apack,
                              int64_t n16, int64_t KS) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = n16 * n16;
    if (idx >= total) return;
    int l = (int)(idx & 63);
    int64_t t = idx >> 6;
    int64_t kk = t % KS, b = t / KS;
    Apack[idx] = P[(16 * b + (l & 15)) * n16 + 4 * kk + (l >> 4)];
}

// pair-packed variant: Apack2[((b*KS/2 + kk2)*64 + l)*2 + t] = fragment element of k-step 2 kk2 + t
__global__ void pack_A2_kernel(const double *__restrict__ P, double *__restrict__ Apack2,
                               int64_t n16, int64_t KS) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = n16 * n16;
    if (idx >= total) return;
    int t = (int)(idx & 1);
    int l = (int)((idx >> 1) & 63);
    int64_t u = idx >> 7;
    int64_t kk2 = u % (KS / 2), b = u / (KS / 2);
    Apack2[idx] = P[(16 * b + (l & 15)) * n16 + 4 * (2 * kk2 + t) + (l >> 4)];
}

// host layout: X[r*n + j]; tile layout: Xt[(r/16) * n16*16 + j*16 + (r%16)]
__global__ void to_tiles_kernel(const double *__restrict__ X, double *__restrict__ Xt, int64_t n,
                                int64_t n16, int64_t R, int64_t Rpad) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Rpad * n16) return;
    int64_t tile = idx / (n16 * 16), rem = idx % (n16 * 16);
    int64_t j = rem >> 4, r = tile * 16 + (rem & 15);
    Xt[idx] = (r < R && j < n) ? X[r * n + j] : 0.0;
}

__global__ void from_tiles_kernel(const double *__restrict__ Xt, double *__restrict__ X,
                                  int64_t n, int64_t n16, int64_t R) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * n) return;
    int64_t r = idx / n, j = idx % n;
    X[idx] = Xt[(r >> 4) * n16 * 16 + j * 16 + (r & 15)];
}

__global__ void randn_tiles_kernel(double *__restrict__ Xt, int64_t n, int64_t n16, int64_t R,
                                   int64_t Rpad, uint64_t seed, uint64_t first_index) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Rpad * n16) return;
    int64_t tile = idx / (n16 * 16), rem = idx % (n16 * 16);
    int64_t j = rem >> 4, r = tile * 16 + (rem & 15);
    Xt[idx] = (r < R && j < n) ? keyed_normal(seed, first_index + (uint64_t)r, (uint64_t)j) : 0.0;
}

// ------------------------------------------------------------------------ MFMA building block

// acc(16 rows of block b) x (16 candidates) += Apack[b][kk0..kk1) * Xt rows.  XT may point to LDS
// or global memory; rows are 16 doubles.
template <typename XPtr>
__device__ inline v4d block_rows_times_X(const double *__restrict__ Ab, XPtr Xs, int kk0, int kk1,
                                         int lane, v4d acc) {
    const int xoff = (lane >> 4) * 16 + (lane & 15);
    int kk = kk0;
    for (; kk + 8 <= kk1; kk += 8) {
        double a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            a[u] = Ab[(int64_t)(kk + u) * 64 + lane];
            b[u] = Xs[(kk + u) * 64 + xoff];
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    for (; kk < kk1; kk++) {
        double a = Ab[(int64_t)kk * 64 + lane];
        double b = Xs[kk * 64 + xoff];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}

// X = mu 1^T + F * Xi   with F packed like P0 (Apack layout) and Xi tile-major.
// One workgroup per tile; wave w computes row blocks b = w, w+4, ...
__global__ __launch_bounds__(256) void affine_tiles_kernel(const double *__restrict__ Fpack,
                                                           const double *__restrict__ mu,
                                                           const double *__restrict__ Xi,
                                                           double *__restrict__ Xt, int64_t n,
                                                           int64_t n16, int64_t NB, int64_t KS) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *Xs = Xi + (int64_t)blockIdx.x * n16 * 16;
    double *Xo = Xt + (int64_t)blockIdx.x * n16 * 16;
    for (int64_t b = wave; b < NB; b += 4) {
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        acc = block_rows_times_X(Fpack + b * KS * 64, Xs, 0, (int)KS, lane, acc);
#pragma unroll
        for (int v = 0; v < 4; v++) {
            int64_t i = 16 * b + (lane >> 4) + 4 * v;
            Xo[i * 16 + (lane & 15)] = (i < n) ? mu[i] + acc[v] : 0.0;
        }
    }
}

// ----------------------------------------------------------------------------------- evaluation

// One workgroup (4 waves) per tile of 16 candidates.
//   objective: y = P0 x by MFMA (wave w owns row blocks w, w+4, ...), f0 = sum_i x_i (y_i + q_i) + r
//   constraints: separable -> element-wise; general -> COO quadratic forms
__global__ __launch_bounds__(256) void eval_kernel(EvalArgs a) {
    __shared__ double red[256];
    __shared__ double redv[256];
    const DevProblem &P = a.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t tile = blockIdx.x;
    const double *Xs = a.X + tile * P.n16 * 16;
    const int r = lane & 15;

    // Four row blocks at a time per wave: the X fragment (B operand) of a k-step is loaded once and
    // feeds four MFMAs on four independent accumulators (dependent back-to-back MFMAs with loads in
    // between run at ~90-140 cycles instead of 64).
    double facc = 0.0;
    const int xoff = (lane >> 4) * 16 + (lane & 15);
    if (a.planes) {
        // quadratic part from the LDS-tiled GEMM (partial planes, fixed order); linear part here
        for (int64_t i = tid >> 4; i < P.n; i += 16) facc = __builtin_fma(Xs[i * 16 + r], P.q0[i], facc);
        if (tid < 16)
            for (int p = 0; p < a.nplanes; p++) facc += a.planes[(int64_t)p * a.Rpad + tile * 16 + tid];
    }
    for (int64_t b0 = 4 * wave; b0 < P.NB && !a.planes; b0 += 16) {
        v4d acc[4];
        const double *Ab[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            acc[u] = v4d{0.0, 0.0, 0.0, 0.0};
            const int64_t bb = (b0 + u < P.NB) ? b0 + u : P.NB - 1;   // tail: recompute the last block, ignored below
            Ab[u] = P.Apack + bb * P.KS * 64 + lane;
        }
        for (int kk = 0; kk < (int)P.KS; kk += 2) {   // KS is a multiple of 4
            const double bx0 = Xs[kk * 64 + xoff], bx1 = Xs[(kk + 1) * 64 + xoff];
            double a0[4], a1[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { a0[u] = Ab[u][(int64_t)kk * 64]; a1[u] = Ab[u][(int64_t)(kk + 1) * 64]; }
#pragma unroll
            for (int u = 0; u < 4; u++) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], bx0, acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; u++) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], bx1, acc[u], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (b0 + u < P.NB) {
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    int64_t i = 16 * (b0 + u) + (lane >> 4) + 4 * v;
                    facc += Xs[i * 16 + r] * (acc[u][v] + P.q0[i]);
                }
            }
        }
    }
    // deterministic reduction over the 16 (wave, lane-group) partials of each candidate
    red[tid] = facc;
    __syncthreads();
    if (tid < 16) {
        double s = 0.0;
        for (int g = 0; g < 16; g++) s += red[g * 16 + tid];
        int64_t gr = tile * 16 + tid;
        double f = s + P.r0;
        a.f0[gr] = f;
        if (a.F) a.F[gr] = f;
    }
    __syncthreads();

    // constraints
    double vmax = -QM_INF;
    const int c = tid >> 4;  // 16 slots x 16 candidates
    const int64_t gr = tile * 16 + (tid & 15);
    if (P.sep) {
        for (int64_t i = c; i < P.n; i += 16) {
            double xi = Xs[i * 16 + (tid & 15)];
            for (int e = P.cptr[i]; e < P.cptr[i + 1]; e++) {
                double f = (P.cp[e] * xi + P.cq[e]) * xi + P.cr[e];
                double v = viol_of(f, P.crel[e]);
                vmax = v > vmax ? v : vmax;
                if (a.F) a.F[(int64_t)P.cidx[e] * a.Rpad + gr] = f;
            }
        }
    } else {
        for (int64_t k = c; k < P.m; k += 16) {
            double s = 0.0;
            for (int64_t e = P.gptr[k]; e < P.gptr[k + 1]; e++)
                s += P.gv[e] * Xs[(int64_t)P.gi[e] * 16 + (tid & 15)] *
                     Xs[(int64_t)P.gj[e] * 16 + (tid & 15)];
            const double *qk = P.gq + k * P.n16;
            double l = 0.0;
            for (int64_t j = 0; j < P.n; j++) l += qk[j] * Xs[j * 16 + (tid & 15)];
            double f = s + l + P.gr[k];
            double v = viol_of(f, P.grel[k]);
            vmax = v > vmax ? v : vmax;
            if (a.F) a.F[(k + 1) * a.Rpad + gr] = f;
        }
    }
    redv[tid] = vmax;
    __syncthreads();
    if (tid < 16) {
        double v = -QM_INF;
        for (int g = 0; g < 16; g++) { double w = redv[g * 16 + tid]; v = w > v ? w : v; }
        a.maxviol[tile * 16 + tid] = v;
    }
}

// -------------------------------------------------------------------- coordinate descent phase 1

// Separable constraints: a coordinate's local violation and its update depend on x_i alone
// (objective is identically zero in phase 1, qcqp.py:114), so a sweep is element-wise; the
// sweep loop, the per-restart max-violation reduction and the termination tests stay in-kernel.
// 1024 threads = 64 coordinate slots x 16 restarts: 4 waves per SIMD hide the latency of the
// long scalar dependency chains (sqrt / divide / Philox) of the bisection.
constexpr int P1_THREADS = 1024, P1_SLOTS = P1_THREADS / 16;
template <int MAXC>
__global__ __launch_bounds__(P1_THREADS) void cd_phase1_sep_kernel(CdArgs a) {
    __shared__ double vred[P1_THREADS];
    __shared__ int ured[P1_THREADS];
    __shared__ double viol_last[16];
    __shared__ int fin[16];
    __shared__ int nlive;
    const DevProblem &P = a.P;
    const int tid = threadIdx.x;
    const int r = tid & 15, slot = tid >> 4;
    const int64_t tile = blockIdx.x;
    double *Xs = a.X + tile * P.n16 * 16;
    const int64_t gr = tile * 16 + r;
    const bool live = gr < a.R;
    if (tid < 16) { viol_last[tid] = QM_INF; fin[tid] = (tile * 16 + tid < a.R) ? 0 : 1; }
    int64_t my_visits = 0, my_acc = 0;
    int my_status = 0;
    int64_t sweeps_done = 0;
    __syncthreads();
    for (int64_t t = 0; t < a.num_iters; t++) {
        // loop-top test of the reference (qcqp.py:111) is folded into fin[]
        if (tid == 0) { int c = 0; for (int k = 0; k < 16; k++) c += fin[k] ? 0 : 1; nlive = c; }
        __syncthreads();
        if (nlive == 0) break;
        const bool run = live && !fin[r];
        double vmax = -QM_INF;
        int upd = 0;
        if (run) {
            if (slot == 0) sweeps_done++;
            for (int64_t i = slot; i < P.n; i += P1_SLOTS) {
                double xi = Xs[i * 16 + r];
                P1Visit V;
                p1_sep_visit<MAXC>(P, i, xi, a.tol, a.viol_tol, a.seed, a.first_index + (uint64_t)gr, t, V);
                if (V.status) my_status = V.status;
                if (!V.visited) continue;
                my_visits++;
                if (V.moved) { Xs[i * 16 + r] = xi; upd = 1; my_acc++; }
                vmax = V.vafter > vmax ? V.vafter : vmax;
            }
        }
        vred[tid] = vmax; ured[tid] = upd;
        __syncthreads();
        if (tid < 16 && !fin[tid]) {
            double v = -QM_INF;
            int u = 0;
            for (int g = 0; g < P1_SLOTS; g++) { double w = vred[g * 16 + tid]; v = w > v ? w : v; u |= ured[g * 16 + tid]; }
            viol_last[tid] = v;
            // done when feasible enough (qcqp.py:111); a sweep without any update is a fixed
            // point of the (deterministic-in-feasibility) map, so later sweeps cannot change x.
            if (v < a.viol_tol || !u) fin[tid] = 1;
        }
        __syncthreads();
    }
    // per-restart outputs
    vred[tid] = (double)my_visits; ured[tid] = (int)my_acc;
    __shared__ int sred[P1_THREADS];
    sred[tid] = my_status;
    __syncthreads();
    if (tid < 16 && tile * 16 + tid < a.R) {
        int64_t vis = 0, acc = 0;
        int st = 0;
        for (int g = 0; g < P1_SLOTS; g++) { vis += (int64_t)vred[g * 16 + tid]; acc += ured[g * 16 + tid]; if (sred[g * 16 + tid]) st = sred[g * 16 + tid]; }
        int64_t g = tile * 16 + tid;
        a.visits[g] = vis; a.accepted[g] = acc; a.status[g] = st;
        a.flag[g] = (viol_last[tid] < a.viol_tol) ? 1 : 0;
    }
    if (slot == 0 && live) a.sweeps[gr] = sweeps_done;
}

// the winners of K populations (select_best_kernel with grid K: out_idx[2 p] = index within population p) -> out[p][0..n)
__global__ void gather_best_x_kernel(const double *__restrict__ Xt, int64_t n, int64_t n16, int64_t R, const int64_t *__restrict__ idx,
                                     double *__restrict__ out) {
    const int64_t p = blockIdx.x, w = idx[2 * p];
    if (w < 0) return;
    const int64_t r = p * R + w;
    const double *src = Xt + (r >> 4) * n16 * 16 + (r & 15);
    for (int64_t j = threadIdx.x; j < n; j += blockDim.x) out[p * n + j] = src[j * 16];
}

// coordinate descent phase 2: cd_phase2.h (general kernel), cd_phase2_rs.h / cd_phase2_q.h (role-split pipelined kernels)

}  // namespace qcqpmi
#include "cd_phase2.h"
#include "cd_phase2_rs.h"
#include "cd_phase2_q.h"
namespace qcqpmi {

// gate of improve_coord_descent (qcqp.py:189): phase 2 runs only for restarts whose max violation
// is below viol_tol (and whose phase 1 did not hit a case where the reference raises)
__global__ void gate_kernel(const double *__restrict__ maxviol, const int *__restrict__ status1,
                            uint8_t *__restrict__ flag, int64_t R, int64_t Rpad, double viol_tol) {
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= Rpad) return;
    flag[r] = (r < R && status1[r] == 0 && maxviol[r] < viol_tol) ? 1 : 0;
}

// ------------------------------------------------------------------------------- best selection

// key = (int(maxviol / tol), f0, index): lexicographic minimum, ties -> lowest index.
__global__ __launch_bounds__(1024) void select_best_kernel(const double *__restrict__ f0,
                                                           const double *__restrict__ maxviol,
                                                           int64_t R, double tol,
                                                           int64_t *__restrict__ out_idx,
                                                           double *__restrict__ out_key) {
    // one workgroup per population: population blockIdx.x is the slice [blockIdx.x R, (blockIdx.x + 1) R) (grid 1: the whole array)
    f0 += (int64_t)blockIdx.x * R; maxviol += (int64_t)blockIdx.x * R;
    out_idx += 2 * (int64_t)blockIdx.x; out_key += 2 * (int64_t)blockIdx.x;
    __shared__ long long sb[1024];
    __shared__ double sf[1024];
    __shared__ long long si[1024];
    long long bb = 0x7fffffffffffffffll, bi = -1;
    double bf = QM_INF;
    for (int64_t r = threadIdx.x; r < R; r += 1024) {
        double v = maxviol[r], f = f0[r];
        long long bucket = (v == v && f == f) ? (long long)(v / tol) : 0x7ffffffffffffffell;
        bool better = bucket < bb || (bucket == bb && f < bf);
        if (bi < 0 || better) { bb = bucket; bf = f; bi = r; }
    }
    sb[threadIdx.x] = bb; sf[threadIdx.x] = bf; si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            int o = threadIdx.x + s;
            bool take = si[o] >= 0 &&
                        (si[threadIdx.x] < 0 || sb[o] < sb[threadIdx.x] ||
                         (sb[o] == sb[threadIdx.x] &&
                          (sf[o] < sf[threadIdx.x] || (sf[o] == sf[threadIdx.x] && si[o] < si[threadIdx.x]))));
            if (take) { sb[threadIdx.x] = sb[o]; sf[threadIdx.x] = sf[o]; si[threadIdx.x] = si[o]; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out_idx[0] = si[0];
        out_idx[1] = sb[0];
        out_key[0] = sf[0];
        out_key[1] = si[0] >= 0 ? maxviol[si[0]] : QM_NAN;
    }
}

}  // namespace qcqpmi
