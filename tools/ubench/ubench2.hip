// MFMA f64 issue-pattern microbenchmarks (what slows v_mfma_f64_16x16x4_f64 down from 64 cycles?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

// mode 0: constant operands, 4 accumulators
// mode 1: B operand rewritten by a VALU move after every MFMA (register recycling)
// mode 2: B operand read from LDS every MFMA (one ahead)
// mode 3: B operand from LDS, distinct registers, 4 accumulators
template <int MODE>
__global__ void k(double *out, long long *cyc, int iters) {
    __shared__ double lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 1e-3;
    __syncthreads();
    v4d acc[4];
    for (int i = 0; i < 4; i++) acc[i] = v4d{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4, b2 = 2.0;
    const int lane = threadIdx.x & 63;
    double breg[16], gacc = 0;
    for (int u = 0; u < 16; u++) breg[u] = lds[u * 64 + lane];
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 16; u++) acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u & 3], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 16; u++) {
                acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u & 3], 0, 0, 0);
                asm volatile("v_mov_b64 %0, %1" : "=v"(b) : "v"(b2));
            }
        } else if (MODE == 2) {
            double bn = lds[(it * 16) % 4032 + lane];
#pragma unroll
            for (int u = 0; u < 16; u++) {
                double bc = bn;
                bn = lds[((it * 16 + u + 1) * 64) % 4032 + lane];
                acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bc, acc[u & 3], 0, 0, 0);
            }
        } else if (MODE == 4) {   // 16 distinct B registers loaded once, no LDS traffic in the loop
#pragma unroll
            for (int u = 0; u < 16; u++) acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, breg[u], acc[u & 3], 0, 0, 0);
        } else if (MODE == 5) {   // distinct A and B registers
#pragma unroll
            for (int u = 0; u < 16; u++) acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(breg[15 - u], breg[u], acc[u & 3], 0, 0, 0);
        } else if (MODE == 6) {   // 16 accumulators-free pattern: same as 4 but a global load in between
#pragma unroll
            for (int u = 0; u < 16; u++) {
                acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, breg[u], acc[u & 3], 0, 0, 0);
                if ((u & 3) == 3) gacc += out[(it * 4 + (u >> 2)) * 64 + lane];
            }
        } else {
            double bb[16];
#pragma unroll
            for (int u = 0; u < 16; u++) bb[u] = lds[((it * 16 + u) * 64) % 4032 + lane];
#pragma unroll
            for (int u = 0; u < 16; u++) acc[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb[u], acc[u & 3], 0, 0, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x + (1 << 20)] = s + b + gacc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    double *out; long long *cyc;
    (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&cyc, 1 << 16);
    long long h[1024];
    const int iters = 1000;
#define RUN(M, B, T) do { hipLaunchKernelGGL((k<M>), dim3(B), dim3(T), 0, 0, out, cyc, iters); (void)hipDeviceSynchronize(); \
    (void)hipMemcpy(h, cyc, sizeof(long long) * B, hipMemcpyDeviceToHost); double avg = 0; for (int i = 0; i < B; i++) avg += h[i]; \
    printf("mode %d blocks %4d threads %4d: %.1f ticks per MFMA\n", M, B, T, avg / B / (iters * 16.0)); } while (0)
    RUN(0, 1, 64); RUN(1, 1, 64); RUN(2, 1, 64); RUN(3, 1, 64);
    RUN(0, 256, 256); RUN(1, 256, 256); RUN(2, 256, 256); RUN(3, 256, 256);
    RUN(4, 1, 64); RUN(5, 1, 64); RUN(6, 1, 64); RUN(4, 256, 256); RUN(5, 256, 256); RUN(6, 256, 256);
    return 0;
}
