// Dependent-latency of mixed fp64 VALU chains on one wave (the chain wave of cd_phase2_rs_kernel is a
// dependent sequence fma -> max -> min -> bfi -> add -> cmp -> cndmask -> fma per coordinate).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s\n", hipGetErrorString(e_)); return 1; } } while (0)
template <int KIND>
__global__ void chain(double *out, long long *ticks, int iters) {
    double x = threadIdx.x * 1e-3 + 0.5, a = 0.25, b = 4.0, t = 1.0001, acc = 0.0;
    const double tol = 1e-9;
    long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (KIND == 0) { x = __builtin_fma(x, t, a); }                                                  // 1 op
            if (KIND == 1) { x = __builtin_fma(x, t, a); x = fmax(x, a); }                                  // 2 ops
            if (KIND == 2) { x = __builtin_fma(x, t, a); x = fmax(x, a); x = fmin(x, b); }                  // 3 ops
            if (KIND == 3) { x = __builtin_fma(x, t, a); x = fmax(x, a); x = fmin(x, b); x = x - a; }       // 4 ops
            if (KIND == 4) { double y = __builtin_fma(x, t, a); y = fmax(y, a); y = fmin(y, b); double d = y - x;
                             x = (fabs(d) > tol) ? y : x; }                                                 // fma,max,min,sub,cmp,cndmask
            if (KIND == 5) { double y = __builtin_fma(x, t, a); y = __builtin_copysign(fmin(fmax(fabs(y), a), b), y);
                             double d = y - x; double dl = (fabs(d) > tol) ? d : 0.0; acc = __builtin_fma(dl, t, acc); x = x + dl; }
        }
    }
    long long t1 = (long long)__builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x + acc;
    if (threadIdx.x == 0) *ticks = t1 - t0;
}
int main(int argc, char **argv) {
    const int NT = argc > 1 ? atoi(argv[1]) : 64;
    double *out; long long *tk, h;
    CHK(hipMalloc(&out, 4096)); CHK(hipMalloc(&tk, 8));
    const int iters = 20000;
    const char *names[] = {"fma", "fma,max", "fma,max,min", "fma,max,min,sub", "fma,max,min,sub,cmp,cndmask x2", "fma,|max|,min,bfi,sub,cmp,cndmask,fma+add"};
    const int nops[] = {1, 2, 3, 4, 6, 8};
#define RUN(K) hipLaunchKernelGGL(chain<K>, dim3(1), dim3(NT), 0, 0, out, tk, iters); CHK(hipDeviceSynchronize()); \
    hipLaunchKernelGGL(chain<K>, dim3(1), dim3(NT), 0, 0, out, tk, iters); CHK(hipDeviceSynchronize()); \
    CHK(hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost)); \
    printf("%-46s %7.1f ticks per link, %5.1f per op\n", names[K], (double)h / (16.0 * iters), (double)h / (16.0 * iters) / nops[K]);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    return 0;
}
