// Round 5 questions about the box, answered before a kernel is designed around the answers:
//  (1) the sustained fp64 MFMA rate against the LENGTH of a launch (1 .. 60 ms, idle gaps between launches): where between
//      the 78 TFLOP/s burst and the 47 TFLOP/s steady state does a 45 ms launch sit?
//  (2) where do the waves of 256-thread workgroups land when two workgroups share a CU (HW_ID: SIMD, CU, SE)?
//  (3) the rate of a launch shaped like a two-workgroups-per-CU chain kernel: per workgroup wave 0 = a latency-bound fp64
//      chain plus a few MFMAs, waves 1-3 stream MFMAs (SIMDs 1-3 carry two streams each, SIMD 0 two chains)
// build: hipcc --offload-arch=gfx950 -O3 -o ubench5 ubench5.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_busy(double *out, int iters) {
    v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-9, y = 1.0 + x;
    for (int i = 0; i < iters; i++) {
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}

__global__ __launch_bounds__(256) void where(unsigned *out) {
    __shared__ double pad[9 * 1024];      // 72 KB: at most two workgroups per CU
    pad[threadIdx.x] = 0.0;
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
    __builtin_amdgcn_s_sleep(127);
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
    // keep the workgroup alive long enough for the whole grid to be resident together
    for (int i = 0; i < 200; i++) __builtin_amdgcn_s_sleep(127);
    if (pad[threadIdx.x] != 0.0) out[0] = 0;
}

// chain-shaped launch: wave 0: `steps` dependent fp64 fmas (x 10 per step) then `cm` MFMAs, per block; waves 1..3: `um` MFMAs per block
__global__ __launch_bounds__(256) void chainlike(double *out, int blocks, int cm, int um, long long *ticks) {
    __shared__ double pad[9 * 1024];
    pad[threadIdx.x] = 0.0;
    const int wave = threadIdx.x >> 6;
    v4d a0 = {0, 0, 0, 0}, a1 = a0;
    double x = threadIdx.x * 1e-9, y = 1.0 + x, z = 0.5;
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    if (wave == 0) {
        for (int b = 0; b < blocks; b++) {
            for (int s = 0; s < 160; s++) z = __builtin_fma(z, 0.999999, x);
            for (int i = 0; i < cm; i += 2) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, z, a1, 0, 0, 0);
            }
        }
    } else {
        for (int b = 0; b < blocks; b++)
            for (int i = 0; i < um; i += 2) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + z + pad[threadIdx.x];
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) ticks[wave] = t1 - t0;
}

int main(int argc, char **argv) {
    double *out;
    hipMalloc(&out, 64 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    // ---- (2) placement
    {
        unsigned *w, hw[512 * 4 * 2];
        hipMalloc(&w, sizeof(hw));
        hipLaunchKernelGGL(where, dim3(512), dim3(256), 0, 0, w);
        hipMemcpy(hw, w, sizeof(hw), hipMemcpyDeviceToHost);
        int bad = 0, hist[4][4] = {{0}};
        for (int b = 0; b < 512; b++)
            for (int wv = 0; wv < 4; wv++) {
                const unsigned h = hw[(b * 4 + wv) * 2];
                const int simd = (h >> 4) & 3;
                hist[wv][simd]++;
                if (simd != wv) bad++;
            }
        printf("placement: 512 workgroups of 4 waves, 72 KB LDS each: waves with SIMD != wave index: %d\n", bad);
        for (int wv = 0; wv < 4; wv++) printf("  wave %d on SIMD 0..3: %d %d %d %d\n", wv, hist[wv][0], hist[wv][1], hist[wv][2], hist[wv][3]);
        // how many workgroups share a CU: count (xcc, se, cu) triples
        int cnt[8][8][16] = {{{0}}};
        for (int b = 0; b < 512; b++) {
            const unsigned h = hw[(b * 4) * 2], x = hw[(b * 4) * 2 + 1];
            cnt[x & 7][(h >> 13) & 7][(h >> 8) & 15]++;
        }
        int h3[8] = {0};
        for (int a = 0; a < 8; a++) for (int s = 0; s < 8; s++) for (int c = 0; c < 16; c++) h3[cnt[a][s][c] > 7 ? 7 : cnt[a][s][c]]++;
        printf("  CUs holding 0,1,2,3,4+ workgroups: %d %d %d %d %d\n", h3[0], h3[1], h3[2], h3[3], h3[4] + h3[5] + h3[6] + h3[7]);
        for (int b = 0; b < 6; b++) printf("  wg %d: hw_id %08x xcc %u  (simd of waves: %u %u %u %u)\n", b, hw[b * 8], hw[b * 8 + 1] & 15,
                                           (hw[b * 8] >> 4) & 3, (hw[b * 8 + 2] >> 4) & 3, (hw[b * 8 + 4] >> 4) & 3, (hw[b * 8 + 6] >> 4) & 3);
    }
    // ---- (1) rate against launch length, 2 waves per SIMD (512 blocks of 4 waves), 20 ms idle before every launch
    {
        const int blocks = 512;
        const int lens[] = {400, 1000, 2500, 5000, 10000, 20000, 40000, 60000};
        for (int rep = 0; rep < 2; rep++)
            for (int li = 0; li < 8; li++) {
                usleep(20000);
                const int iters = lens[li];
                hipEventRecord(e0);
                hipLaunchKernelGGL(mfma_busy, dim3(blocks), dim3(256), 0, 0, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                const double fl = (double)blocks * 4 * (double)iters * 4 * 2048;
                printf("burst: iters %6d  %7.2f ms  %6.2f TFLOP/s\n", iters, ms, fl / ms / 1e9);
            }
        // back to back 45 ms launches without gaps
        for (int k = 0; k < 6; k++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_busy, dim3(blocks), dim3(256), 0, 0, out, 45000);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("back-to-back: %7.2f ms  %6.2f TFLOP/s\n", ms, (double)blocks * 4 * 45000.0 * 4 * 2048 / ms / 1e9);
        }
    }
    // ---- (3) chain-shaped launches
    {
        long long *ticks, ht[4];
        hipMalloc(&ticks, 32);
        struct { int wgs, cm, um; } cfg[] = {{256, 24, 160}, {512, 24, 80}, {512, 40, 76}, {512, 56, 72}, {512, 0, 80}, {512, 80, 80}, {256, 80, 80}};
        for (auto c : cfg)
            for (int rep = 0; rep < 2; rep++) {
                usleep(20000);
                const int blocks = 4000;
                hipEventRecord(e0);
                hipLaunchKernelGGL(chainlike, dim3(c.wgs), dim3(256), 0, 0, out, blocks, c.cm, c.um, ticks);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(ht, ticks, 32, hipMemcpyDeviceToHost);
                const double fl = (double)c.wgs * blocks * (c.cm + 3.0 * c.um) * 2048;
                printf("chainlike: %3d wgs, chain %2d + 3 x %3d MFMAs per block: %7.2f ms  %6.2f TFLOP/s issued; ticks per block: chain wave %.0f, mfma wave %.0f\n",
                       c.wgs, c.cm, c.um, ms, fl / ms / 1e9, (double)ht[0] / blocks, (double)ht[1] / blocks);
            }
    }
    return 0;
}
