// Can a wave get a few double-precision VALU adds through while the other wave of its SIMD streams
// v_mfma_f64_16x16x4_f64 back to back?  Wave 1 streams MFMAs; wave 5 (same SIMD) waits a little, then times 8 v_add_f64
// (at priority 0 or 3), a burst of fp32 adds and LDS stores for comparison.
// build: hipcc --offload-arch=gfx950 -O3 -o dp_contention dp_contention.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int PRIO, int STREAM_PRIO>
__global__ __launch_bounds__(512) void k(double *out, long long *cyc, int iters) {
    __shared__ double buf[1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (wave == 1) {
        __builtin_amdgcn_s_setprio(STREAM_PRIO);
        v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        double x = 1.0 + lane, y = 0.5;
        long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; it++) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        }
        long long t1 = __builtin_amdgcn_s_memtime();
        a0 = (a0 + a1) + (a2 + a3);
        out[tid] = a0[0] + a0[1] + a0[2] + a0[3];
        if (lane == 0) { cyc[0] = t1 - t0; }
    } else if (wave == 5) {
        __builtin_amdgcn_s_sleep(100);
        __builtin_amdgcn_s_setprio(PRIO);
        double d[8];
        for (int i = 0; i < 8; i++) d[i] = 1.0 + lane + i;
        float f[8];
        for (int i = 0; i < 8; i++) f[i] = 1.0f + lane + i;
        asm volatile("" ::: "memory");
        long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
        asm volatile("s_nop 0" ::: "memory");
        long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
        asm volatile("s_nop 0" ::: "memory");
        long long t2 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 4; i++) buf[i * 64 + lane] = d[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        long long t3 = __builtin_amdgcn_s_memtime();
        double rd[4];
#pragma unroll
        for (int i = 0; i < 4; i++) rd[i] = *(volatile double *)&buf[i * 64 + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        long long t4 = __builtin_amdgcn_s_memtime();
        const double *gp = out + 4096 + lane;
        double gv = *(volatile const double *)gp;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long t5 = __builtin_amdgcn_s_memtime();
        d[0] += rd[0] + rd[1] + rd[2] + rd[3] + gv;
        double s = 0; float sf = 0;
        for (int i = 0; i < 8; i++) { s += d[i]; sf += f[i]; }
        out[tid] = s + sf + buf[lane];
        if (lane == 0) { cyc[1] = t1 - t0; cyc[2] = t2 - t1; cyc[3] = t3 - t2; cyc[4] = t4 - t3; cyc[5] = t5 - t4; }
    }
}

int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 8192 * 8); hipMemset(out, 0, 8192 * 8); hipMalloc(&cyc, 64);
    struct { const char *n; void (*f)(double *, long long *, int); } v[] = {
        {"adds prio 0, stream prio 0", k<0, 0>}, {"adds prio 3, stream prio 0", k<3, 0>},
        {"adds prio 3, stream prio 2", k<3, 2>}, {"adds prio 0, stream prio 2", k<0, 2>}};
    for (auto &e : v)
        for (int iters : {0, 2000}) {
            hipMemset(cyc, 0, 64);
            hipLaunchKernelGGL(e.f, dim3(1), dim3(512), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
            long long h[6]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
            printf("%-28s %s: stream %lld cycles; 8 v_add_f64 %lld, 8 v_add_f32 %lld, 4 ds_write_b64 %lld, 4 ds_read_b64 %lld, 1 global load %lld cycles\n", e.n,
                   iters ? "partner streaming MFMAs" : "partner idle           ", h[0], h[1], h[2], h[3], h[4], h[5]);
        }
    return 0;
}
