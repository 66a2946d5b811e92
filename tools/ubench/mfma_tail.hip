// Two waves of one SIMD stream v_mfma_f64_16x16x4_f64; the younger one (wave 5) stops after `iters` MFMAs and adds its
// accumulators.  How long after its last MFMA was ISSUED do the results arrive while the elder keeps streaming?
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_tail mfma_tail.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int P1, int P5, int P5END>
__global__ __launch_bounds__(512) void k(double *out, long long *cyc, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave != 1 && wave != 5) return;
    if (wave == 1) __builtin_amdgcn_s_setprio(P1); else __builtin_amdgcn_s_setprio(P5);
    v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = 1.0 + lane, y = 0.5;
    const int n = wave == 1 ? 3 * iters : iters;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (wave == 1) {
        for (int it = 0; it < n / 2; it++) {   // the elder: four independent chains, back to back
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        }
        a0 += a2; a1 += a3;
    } else {
        for (int it = 0; it < n; it++) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (wave == 5) __builtin_amdgcn_s_setprio(P5END);
    asm volatile("" ::: "memory");
    a0 = a0 + a1;
    asm volatile("s_nop 0" : "+v"(a0) :: "memory");
    long long t2 = __builtin_amdgcn_s_memtime();
    out[tid] = a0[0] + a0[1] + a0[2] + a0[3];
    if (lane == 0) { cyc[wave * 4] = t0; cyc[wave * 4 + 1] = t1; cyc[wave * 4 + 2] = t2; }
}

int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 512 * 8); hipMalloc(&cyc, 64 * 8);
    struct { const char *n; void (*f)(double *, long long *, int); } v[] = {
        {"elder prio 0, younger 0, end 0", k<0, 0, 0>}, {"elder prio 0, younger 0, end 3", k<0, 0, 3>},
        {"elder prio 0, younger 2, end 3", k<0, 2, 3>}, {"elder prio 2, younger 0, end 3", k<2, 0, 3>}};
    for (auto &e : v) {
        const int iters = 40;
        hipMemset(cyc, 0, 64 * 8);
        hipLaunchKernelGGL(e.f, dim3(1), dim3(512), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        long long h[64]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        long long b = h[4];
        printf("%-32s elder: %d MFMAs issued over [%lld, %lld], added by %lld | younger: %d MFMAs issued over [%lld, %lld], added by %lld (+%lld after the last issue)\n",
               e.n, 6 * iters, h[4] - b, h[5] - b, h[6] - b, 2 * iters, h[20] - b, h[21] - b, h[22] - b, h[22] - h[21]);
    }
    return 0;
}
