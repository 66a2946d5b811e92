// Micro-benchmarks that fix the constants DESIGN.md quotes: fp64 VALU / MFMA issue and latency on
// gfx950, and the s_memtime tick rate.  Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int ILP>
__global__ void fma_chain(double *out, long long *cyc, int iters) {
    double a[ILP];
    for (int i = 0; i < ILP; i++) a[i] = threadIdx.x * 1e-3 + i;
    double m = 1.0000001, c = 1e-9;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int i = 0; i < ILP; i++) a[i] = __builtin_fma(a[i], m, c);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < ILP; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int ILP>
__global__ void mfma_chain(double *out, long long *cyc, int iters) {
    v4d acc[ILP];
    for (int i = 0; i < ILP; i++) acc[i] = v4d{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < ILP; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < ILP; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void tick_rate(long long *cyc, int spin) {
    long long t0 = __builtin_amdgcn_s_memtime();
    long long t = t0;
    while (t - t0 < spin) t = __builtin_amdgcn_s_memtime();
    cyc[0] = t - t0;
}

int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 1 << 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    long long h[4096];
    auto report = [&](const char *name, int blocks, int threads, double ops_per_wave, float ms) {
        hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < blocks; i++) avg += h[i]; avg /= blocks;
        printf("%-34s blocks=%5d thr=%4d  ticks/op=%8.2f  wall=%8.3f ms\n", name, blocks, threads, avg / ops_per_wave, ms);
    };
    float ms;
    // tick rate
    hipEventRecord(e0); hipLaunchKernelGGL(tick_rate, dim3(1), dim3(64), 0, 0, cyc, 100000000); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("s_memtime: %lld ticks in %.3f ms => %.1f MHz\n", h[0], ms, h[0] / ms / 1e3);
    const int iters = 2000;
#define RUN(K, ILP, B, T, OPS) do { hipLaunchKernelGGL((K<ILP>), dim3(B), dim3(T), 0, 0, out, cyc, iters); hipDeviceSynchronize(); \
    hipEventRecord(e0); hipLaunchKernelGGL((K<ILP>), dim3(B), dim3(T), 0, 0, out, cyc, iters); hipEventRecord(e1); hipEventSynchronize(e1); \
    hipEventElapsedTime(&ms, e0, e1); report(#K " ILP=" #ILP, B, T, (double)iters * OPS * ILP, ms); \
    if (B >= 256) printf("    -> %.2f TFLOP/s\n", (double)B * (T / 64) * iters * OPS * ILP * FL / (ms * 1e-3) / 1e12); } while (0)
    { const double FL = 128;  // flops per wave-level v_fma_f64
      RUN(fma_chain, 1, 1, 64, 16); RUN(fma_chain, 2, 1, 64, 16); RUN(fma_chain, 4, 1, 64, 16); RUN(fma_chain, 8, 1, 64, 16);
      RUN(fma_chain, 8, 1024, 256, 16); RUN(fma_chain, 8, 2048, 256, 16); RUN(fma_chain, 4, 4096, 256, 16); }
    { const double FL = 2048;  // flops per v_mfma_f64_16x16x4_f64
      RUN(mfma_chain, 1, 1, 64, 8); RUN(mfma_chain, 2, 1, 64, 8); RUN(mfma_chain, 4, 1, 64, 8);
      RUN(mfma_chain, 4, 256, 256, 8); RUN(mfma_chain, 4, 1024, 256, 8); RUN(mfma_chain, 4, 2048, 256, 8); }
    return 0;
}
