// Which CUs does a stream created with hipExtStreamCreateWithCUMask use on an MI355X (8 XCCs x 32 CUs)?  Every workgroup
// records (XCC_ID, HW_ID) and holds its CU for a while (large LDS: one workgroup per CU); the host counts distinct CUs per XCC.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <set>
#include <map>
__global__ void probe(unsigned *out, int spin) {
    extern __shared__ double lds[];
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; lds[0] = 1.0; }
    long long t0 = clock64();
    while (clock64() - t0 < (long long)spin) {}
}
static void run(const char *tag, hipStream_t st, int wgs) {
    unsigned *d; hipMalloc(&d, wgs * 2 * sizeof(unsigned));
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(probe, dim3(wgs), dim3(512), 150 * 1024, st, d, 2000000);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(wgs * 2);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per;
    for (int i = 0; i < wgs; i++) per[h[2 * i] & 0xf].insert((h[2 * i + 1] >> 8) & 0xfff);   // HW_ID: CU id [11:8], SH [12], SE [15:13]
    printf("%s: %d workgroups ->", tag, wgs);
    int tot = 0;
    for (auto &kv : per) { printf(" xcc%u:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
    printf("  total distinct CUs %d\n", tot);
    hipFree(d);
}
int main() {
    hipStream_t s0; hipStreamCreate(&s0);
    run("unmasked", s0, 256);
    {   // 256-bit mask, 24 of every 32 bits
        uint32_t m[8]; for (int i = 0; i < 8; i++) m[i] = 0x00ffffffu;
        hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m);
        printf("mask 8 words x 0x00ffffff: %s\n", hipGetErrorString(e));
        if (e == hipSuccess) run("8x24", s, 256);
    }
    {   // only the first word
        uint32_t m[1] = {0x00ffffffu};
        hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, 1, m);
        printf("mask 1 word 0x00ffffff: %s\n", hipGetErrorString(e));
        if (e == hipSuccess) run("1x24", s, 256);
    }
    {   // first four words full
        uint32_t m[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0};
        hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m);
        printf("mask first 128 bits: %s\n", hipGetErrorString(e));
        if (e == hipSuccess) run("128", s, 256);
    }
    {   // alternate bits
        uint32_t m[8]; for (int i = 0; i < 8; i++) m[i] = 0x55555555u;
        hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m);
        printf("mask 0x55555555 x 8: %s\n", hipGetErrorString(e));
        if (e == hipSuccess) run("alt", s, 256);
    }
    return 0;
}
