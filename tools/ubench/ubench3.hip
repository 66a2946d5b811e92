// Sustained fp64 MFMA rate of the whole chip (power / clock behaviour under a long dense-matrix load):
// back-to-back launches of independent v_mfma_f64_16x16x4_f64 chains, TFLOP/s per ~50 ms window.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_busy(double *out, int iters, long long *ticks) {
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-9, y = 1.0 + x;
    for (int i = 0; i < iters; i++) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticks = (long long)__builtin_amdgcn_s_memtime() - t0;
}
int main(int argc, char **argv) {
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 1;
    double *out;
    hipMalloc(&out, 8 << 20);
    long long *ticks, ht = 0;
    hipMalloc(&ticks, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd, iters = 50000;   // per launch: blocks*4 waves * iters*4 MFMAs
    for (int w = 0; w < 30; w++) {
        hipEventRecord(e0);
        for (int k = 0; k < 4; k++) hipLaunchKernelGGL(mfma_busy, dim3(blocks), dim3(256), 0, 0, out, iters, ticks);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double fl = 4.0 * blocks * 4 * (double)iters * 4 * 2048;
        hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost);
        printf("window %2d: %7.2f ms  %6.2f TFLOP/s   s_memtime ticks per MFMA of wave 0: %.1f (kernel %.2f ms => %.0f MHz tick rate)\n", w, ms, fl / ms / 1e9, (double)ht / (4.0 * iters), ms / 4, ht / (ms / 4) / 1e3);
    }
    return 0;
}
