// General fp64 GEMM on v_mfma_f64_16x16x4_f64 for operands that already live in the engine's native layouts
// (no transposes, no vendor BLAS):
//
//     C[t][i][16] = bias[i] + sum_k A[i][k] B[t][k][16]          t = tile of 16 columns
//
//   A   fragment-packed by blocks of 16 rows:  Apk[rb][kk][l] = A[16 rb + (l & 15)][4 kk + (l >> 4)]
//       (MB row blocks, 4 KB k-steps: K padded to a multiple of 16) -- one coalesced 512-byte wave load per MFMA operand
//   B,C tile-major, 16 columns side by side (the population layout of kernels.h): four consecutive rows of a
//       tile are exactly one MFMA B operand
//
// Workgroup = 8 row blocks x 8 tiles (128 x 128 outputs), 4 waves as 2 x 2, each wave a 4 x 4 register block of
// accumulators; both operands stream through LDS in stages of 4 k-steps (16 contiguous 2-KB chunks per stage,
// 16-byte loads), the loads of stage s + 2 in flight while stage s multiplies (register prefetch + LDS double
// buffer, one barrier per stage) -- the tiling of dense_products_kernel (cd_dense.h), without its special modes.
// The contraction can be split over grid.z into `zs` partial planes C + z * plane (summed by the consumer in a
// fixed order: deterministic) for products with few outputs and a long K (the ADMM consensus sum).
//
// Used by the ADMM improve path (capi_admm.inc): ZQ = W^T Z, S = W D, z = Minv rhs, Y = P0 z.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qcqpmi {

struct GemmPkArgs {
    const double *A;      // [MB][4 KB][64]
    const double *B;      // [ntiles][16 KB * 16]   (rows 16 KB)
    double *C;            // [zs][ntiles][16 MB * 16]
    const double *bias;   // [16 MB] added by plane 0, or nullptr
    int MB, KB, ntiles, zs;
    double alpha;         // C = alpha * (A B) + bias
};

constexpr int GP_FG = 8, GP_TG = 8, GP_KC = 4;
constexpr int GP_LDS_BYTES = 2 * (GP_FG + GP_TG) * GP_KC * 64 * 8;

typedef double gp_v4d __attribute__((ext_vector_type(4)));
typedef double gp_v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256, 2) void gemm_pk_kernel(GemmPkArgs a) {
    extern __shared__ double smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int f0 = GP_FG * blockIdx.x, tg0 = GP_TG * blockIdx.y, z = blockIdx.z;
    const int KS = 4 * a.KB;
    // this thread's slice of the 16 streams: passes 0..3 -> row blocks, 4..7 -> tiles
    const int half = tid >> 7, off = (tid & 127) * 2;
    const double *src[8];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        int f = f0 + 2 * p + half, t = tg0 + 2 * p + half;
        f = f < a.MB ? f : a.MB - 1;            // clamped: loaded, multiplied, never stored
        t = t < a.ntiles ? t : a.ntiles - 1;
        src[p] = a.A + ((int64_t)f * KS) * 64 + off;
        src[4 + p] = a.B + (int64_t)t * a.KB * 256 + off;
    }
    const int rb0 = f0 + 4 * wm, tl0 = tg0 + 4 * wn;   // this wave's row blocks / tiles
    const int ch_lo = (int)((int64_t)z * a.KB / a.zs), ch_hi = (int)((int64_t)(z + 1) * a.KB / a.zs);
    gp_v4d acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
        for (int t = 0; t < 4; t++) acc[u][t] = gp_v4d{0.0, 0.0, 0.0, 0.0};
    if (ch_hi > ch_lo) {
        gp_v2d pf[8];
#pragma unroll
        for (int p = 0; p < 8; p++) pf[p] = *reinterpret_cast<const gp_v2d *>(src[p] + (int64_t)ch_lo * 256);
        int buf = 0;
#pragma unroll
        for (int p = 0; p < 8; p++)
            *reinterpret_cast<gp_v2d *>(smem + ((buf * 16 + 2 * p + half) * 256 + off)) = pf[p];
        {
            const int c1 = (ch_lo + 1 < ch_hi) ? ch_lo + 1 : ch_lo;
#pragma unroll
            for (int p = 0; p < 8; p++) pf[p] = *reinterpret_cast<const gp_v2d *>(src[p] + (int64_t)c1 * 256);
        }
        __syncthreads();
        for (int ch = ch_lo; ch < ch_hi; ch++) {
#pragma unroll
            for (int p = 0; p < 8; p++)
                *reinterpret_cast<gp_v2d *>(smem + (((buf ^ 1) * 16 + 2 * p + half) * 256 + off)) = pf[p];
            const int chn = (ch + 2 < ch_hi) ? ch + 2 : ch_hi - 1;
#pragma unroll
            for (int p = 0; p < 8; p++) pf[p] = *reinterpret_cast<const gp_v2d *>(src[p] + (int64_t)chn * 256);
            const double *As = smem + (buf * 16 + 4 * wm) * 256 + lane;
            const double *Bs = smem + (buf * 16 + 8 + 4 * wn) * 256 + lane;
#pragma unroll
            for (int k2 = 0; k2 < GP_KC; k2 += 2) {
                double av[2][4], bv[2][4];
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
#pragma unroll
                    for (int u = 0; u < 4; u++) { av[ks][u] = As[u * 256 + (k2 + ks) * 64]; bv[ks][u] = Bs[u * 256 + (k2 + ks) * 64]; }
#pragma unroll
                for (int ks = 0; ks < 2; ks++)
#pragma unroll
                    for (int t = 0; t < 4; t++)
#pragma unroll
                        for (int u = 0; u < 4; u++)
                            acc[u][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks][u], bv[ks][t], acc[u][t], 0, 0, 0);
            }
            __syncthreads();
            buf ^= 1;
        }
    }
    // accumulator layout: register v of lane l = row (l >> 4) + 4 v of the block, column l & 15 of the tile
    double *Cz = a.C + (int64_t)z * a.ntiles * a.MB * 256;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        if (rb0 + u >= a.MB) continue;   // wave-uniform
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (tl0 + t >= a.ntiles) continue;
            double *co = Cz + ((int64_t)(tl0 + t) * a.MB + (rb0 + u)) * 256;
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int i = (lane >> 4) + 4 * v;
                const double bi = (a.bias && z == 0) ? a.bias[16 * (int64_t)(rb0 + u) + i] : 0.0;
                co[i * 16 + (lane & 15)] = a.alpha * acc[u][t][v] + bi;
            }
        }
    }
}

// The same product with 64 x 64 outputs per workgroup (4 row blocks x 4 tiles, each wave a 2 x 2 register block): four
// times as many workgroups for the small products of the reduced-basis ADMM iteration (160 x 1024 x 1024 and
// 1024 x 160 x 1024 fill 16 and 64 workgroups of the large tiling on 256 CUs).
constexpr int GS_FG = 4, GS_TG = 4;
constexpr int GS_LDS_BYTES = 2 * (GS_FG + GS_TG) * GP_KC * 64 * 8;

__global__ __launch_bounds__(256, 2) void gemm_pk_small_kernel(GemmPkArgs a) {
    extern __shared__ double smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int f0 = GS_FG * blockIdx.x, tg0 = GS_TG * blockIdx.y, z = blockIdx.z;
    const int KS = 4 * a.KB;
    // this thread's slice of the 8 streams: passes 0..1 -> row blocks, 2..3 -> tiles
    const int half = tid >> 7, off = (tid & 127) * 2;
    const double *src[4];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        int f = f0 + 2 * p + half, t = tg0 + 2 * p + half;
        f = f < a.MB ? f : a.MB - 1;            // clamped: loaded, multiplied, never stored
        t = t < a.ntiles ? t : a.ntiles - 1;
        src[p] = a.A + ((int64_t)f * KS) * 64 + off;
        src[2 + p] = a.B + (int64_t)t * a.KB * 256 + off;
    }
    const int rb0 = f0 + 2 * wm, tl0 = tg0 + 2 * wn;   // this wave's row blocks / tiles
    const int ch_lo = (int)((int64_t)z * a.KB / a.zs), ch_hi = (int)((int64_t)(z + 1) * a.KB / a.zs);
    gp_v4d acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
        for (int t = 0; t < 2; t++) acc[u][t] = gp_v4d{0.0, 0.0, 0.0, 0.0};
    if (ch_hi > ch_lo) {
        gp_v2d pf[4];
#pragma unroll
        for (int p = 0; p < 4; p++) pf[p] = *reinterpret_cast<const gp_v2d *>(src[p] + (int64_t)ch_lo * 256);
        int buf = 0;
#pragma unroll
        for (int p = 0; p < 4; p++)
            *reinterpret_cast<gp_v2d *>(smem + ((buf * 8 + 2 * p + half) * 256 + off)) = pf[p];
        {
            const int c1 = (ch_lo + 1 < ch_hi) ? ch_lo + 1 : ch_lo;
#pragma unroll
            for (int p = 0; p < 4; p++) pf[p] = *reinterpret_cast<const gp_v2d *>(src[p] + (int64_t)c1 * 256);
        }
        __syncthreads();
        for (int ch = ch_lo; ch < ch_hi; ch++) {
#pragma unroll
            for (int p = 0; p < 4; p++)
                *reinterpret_cast<gp_v2d *>(smem + (((buf ^ 1) * 8 + 2 * p + half) * 256 + off)) = pf[p];
            const int chn = (ch + 2 < ch_hi) ? ch + 2 : ch_hi - 1;
#pragma unroll
            for (int p = 0; p < 4; p++) pf[p] = *reinterpret_cast<const gp_v2d *>(src[p] + (int64_t)chn * 256);
            const double *As = smem + (buf * 8 + 2 * wm) * 256 + lane;
            const double *Bs = smem + (buf * 8 + 4 + 2 * wn) * 256 + lane;
#pragma unroll
            for (int k2 = 0; k2 < GP_KC; k2++) {
                double av[2], bv[2];
#pragma unroll
                for (int u = 0; u < 2; u++) { av[u] = As[u * 256 + k2 * 64]; bv[u] = Bs[u * 256 + k2 * 64]; }
#pragma unroll
                for (int t = 0; t < 2; t++)
#pragma unroll
                    for (int u = 0; u < 2; u++)
                        acc[u][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[t], acc[u][t], 0, 0, 0);
            }
            __syncthreads();
            buf ^= 1;
        }
    }
    double *Cz = a.C + (int64_t)z * a.ntiles * a.MB * 256;
#pragma unroll
    for (int u = 0; u < 2; u++) {
        if (rb0 + u >= a.MB) continue;   // wave-uniform
#pragma unroll
        for (int t = 0; t < 2; t++) {
            if (tl0 + t >= a.ntiles) continue;
            double *co = Cz + ((int64_t)(tl0 + t) * a.MB + (rb0 + u)) * 256;
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int i = (lane >> 4) + 4 * v;
                const double bi = (a.bias && z == 0) ? a.bias[16 * (int64_t)(rb0 + u) + i] : 0.0;
                co[i * 16 + (lane & 15)] = a.alpha * acc[u][t][v] + bi;
            }
        }
    }
}

// pack a matrix given by an element functor-free pair of strides into the fragment layout:
//   Apk[rb][kk][l] = src[(16 rb + (l & 15)) * rs + (4 kk + (l >> 4)) * cs]   (zero outside rows x cols)
__global__ void gemm_pk_pack_kernel(const double *__restrict__ src, double *__restrict__ Apk, int64_t rows, int64_t cols,
                                    int64_t rs, int64_t cs, int MB, int KB) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)MB * KB * 256;
    if (idx >= total) return;
    const int l = (int)(idx & 63);
    const int64_t fk = idx >> 6;
    const int64_t rb = fk / (4 * KB), kk = fk % (4 * KB);
    const int64_t row = 16 * rb + (l & 15), col = 4 * kk + (l >> 4);
    Apk[idx] = (row < rows && col < cols) ? src[row * rs + col * cs] : 0.0;
}

}  // namespace qcqpmi
