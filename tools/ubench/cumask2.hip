// Do k workgroups that need a CU each (150 KB of LDS) run CONCURRENTLY on a stream whose CU mask has its first k bits set?
// Every workgroup records (XCC_ID, CU) and waits until all k have arrived or a time limit passes: a persistent kernel of k
// workgroups (the ring mode of cd_phase2_qs_kernel) needs exactly that.  usage: cumask2 [k ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>
#include <map>
__global__ void probe(unsigned *out, int *arrived, int k, long long limit) {
    extern __shared__ double lds[];
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    int seen = 0;
    if (threadIdx.x == 0) {
        lds[0] = 1.0;
        const int mine = atomicAdd(arrived, 1);
        long long t0 = wall_clock64();
        do { seen = __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (seen < k && wall_clock64() - t0 < limit);
        out[4 * blockIdx.x] = xcc; out[4 * blockIdx.x + 1] = hw; out[4 * blockIdx.x + 2] = (unsigned)mine; out[4 * blockIdx.x + 3] = (unsigned)seen;
    }
}
int main(int argc, char **argv) {
    std::vector<int> ks;
    for (int i = 1; i < argc; i++) ks.push_back(atoi(argv[i]));
    if (ks.empty()) ks = {176, 192, 200, 208, 224};
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);     // kHz
    for (int k : ks) {
        uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < k; i++) m[i / 32] |= 1u << (i % 32);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, m) != hipSuccess) { printf("k=%d: mask refused\n", k); continue; }
        unsigned *d; int *arr;
        hipMalloc(&d, k * 4 * sizeof(unsigned)); hipMalloc(&arr, sizeof(int)); hipMemset(arr, 0, sizeof(int));
        hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipLaunchKernelGGL(probe, dim3(k), dim3(512), 150 * 1024, s, d, arr, k, (long long)rate * 200);     // 200 ms
        hipStreamSynchronize(s);
        std::vector<unsigned> h(k * 4);
        hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
        std::map<unsigned, std::set<unsigned>> per;
        std::map<unsigned, int> wgs;
        unsigned minseen = 1u << 30;
        for (int i = 0; i < k; i++) { per[h[4 * i] & 0xf].insert((h[4 * i + 1] >> 8) & 0xfff); wgs[h[4 * i] & 0xf]++; if (h[4 * i + 3] < minseen) minseen = h[4 * i + 3]; }
        printf("first %3d bits, %3d workgroups: all concurrent %s (a workgroup saw at least %u arrive);", k, k, minseen >= (unsigned)k ? "YES" : "NO ", minseen);
        for (auto &kv : per) printf(" xcc%u:%zu CUs/%d wgs", kv.first, kv.second.size(), wgs[kv.first]);
        printf("\n");
        hipFree(d); hipFree(arr); hipStreamDestroy(s);
    }
    return 0;
}
