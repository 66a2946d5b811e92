// Issue-rate microbenchmark for the phase-2 product loop: how many cycles per v_mfma_f64_16x16x4_f64 do one or two
// waves of a SIMD sustain with the loop's companion instructions (LDS operand reads, global fragment loads)?
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_issue mfma_issue.hip ; run: ./mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

template <int MODE>   // bit 0: LDS operand reads, bit 1: global fragment loads (ring of 11 units), bit 2: branch per unit
__global__ __launch_bounds__(512) void k(const double *A, double *out, long long *cyc, int iters, int active_mask) {
    extern __shared__ double X[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 512) X[i] = 1e-3 * i;
    __syncthreads();
    if (!((active_mask >> wave) & 1)) return;
    v4d acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    v2d ring[22];
    const char *Ab = (const char *)A + (size_t)blockIdx.x * 0;
    unsigned voff[22];
#pragma unroll
    for (int u = 0; u < 22; u++) voff[u] = lane * 16 + u * 6144;
#pragma unroll
    for (int u = 0; u < 11; u++) { ring[2 * u] = *(const v2d *)(Ab + voff[u]); ring[2 * u + 1] = *(const v2d *)(Ab + voff[u] + 1024); }
    const double *xb = X + lane + (wave & 3) * 256;
    double bc[4] = {1.0, 2.0, 3.0, 4.0}, bn[4] = {1.0, 2.0, 3.0, 4.0};
    unsigned skip = (unsigned)iters >> 20;   // zero, but the compiler cannot know
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        const char *rp = Ab + (size_t)(it & 63) * 131072;
#pragma unroll
        for (int half = 0; half < 2; half++) {
#pragma unroll
            for (int U = 0; U < 11; U++) {
                const int u = 11 * half + U;
                if (MODE & 1) {
#pragma unroll
                    for (int q = 0; q < 4; q++) bn[q] = xb[(12 * ((u + 1) % 21) + q) * 64];
                }
                if (!(MODE & 4) || !((skip >> u) & 1u)) {
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[2 * U][0], bc[0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[2 * U][1], bc[1], acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[2 * U + 1][0], bc[2], acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[2 * U + 1][1], bc[3], acc3, 0, 0, 0);
                }
                if (MODE & 2) {
                    const unsigned vo = half == 0 ? voff[U + 11] : voff[U];
                    ring[2 * U] = *(const v2d *)(rp + vo);
                    ring[2 * U + 1] = *(const v2d *)(rp + vo + 1024);
                }
                if (MODE & 1) {
#pragma unroll
                    for (int q = 0; q < 4; q++) bc[q] = bn[q];
                }
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    acc0 = (acc0 + acc1) + (acc2 + acc3);
    out[(size_t)blockIdx.x * 512 + tid] = acc0[0] + acc0[1] + acc0[2] + acc0[3];
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    double *A, *out; long long *cyc;
    hipMalloc(&A, 64 * 131072 + 262144); hipMemset(A, 0, 64 * 131072 + 262144);
    hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 2000;
    const char *names[8] = {"mfma only", "+lds", "+global", "+lds+global", "mfma+branch", "+lds+br", "+global+br", "+lds+global+br"};
    struct { const char *n; int mask; } occ[] = {{"1 wave/SIMD (waves 0-3)", 0x0F}, {"2 waves/SIMD (0-7)", 0xFF}, {"1 wave on SIMD1 only", 0x02}, {"2 waves on SIMD1 only", 0x22}};
    for (auto &o : occ)
        for (int mode = 0; mode < 8; mode++) {
            hipMemset(cyc, 0, 256 * 8 * 8);
            void (*kk)(const double *, double *, long long *, int, int) =
                mode == 0 ? k<0> : mode == 1 ? k<1> : mode == 2 ? k<2> : mode == 3 ? k<3> : mode == 4 ? k<4> : mode == 5 ? k<5> : mode == 6 ? k<6> : k<7>;
            hipFuncSetAttribute((const void *)kk, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(kk, dim3(256), dim3(512), 16384 * 8, 0, A, out, cyc, iters, o.mask);   // warm
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kk, dim3(256), dim3(512), 16384 * 8, 0, A, out, cyc, iters, o.mask);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            long long h[8]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
            int nw = __builtin_popcount(o.mask & 0x22 ? (o.mask & 0x22) : o.mask);
            // cycles per MFMA as seen by the SIMD-1 pipe: time of wave 1 / (mfmas of all waves on SIMD 1)
            int on1 = ((o.mask >> 1) & 1) + ((o.mask >> 5) & 1);
            double per = (double)h[1] / (88.0 * iters * on1);
            const double mf = 256.0 * __builtin_popcount(o.mask) * 88.0 * iters;
            printf("%-26s %-16s wave1 ticks %lld  -> %.1f ticks per MFMA on the SIMD; %.3f ms, %.1f TFLOP/s, tick rate %.2f GHz\n", o.n, names[mode], h[1], per, ms, mf * 2048 / ms / 1e9, h[1] / ms / 1e6);
            (void)nw;
        }
    return 0;
}
