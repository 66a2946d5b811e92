// Which CU does bit i of a hipExtStreamCreateWithCUMask mask enable?  One-bit masks, one workgroup each: prints XCC and HW_ID fields.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) { out[0] = xcc; out[1] = hw; }
}
int main() {
    unsigned *d; hipMalloc(&d, 8);
    std::vector<int> xccs(256, -1), cus(256, -1);
    for (int i = 0; i < 256; i++) {
        uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        m[i / 32] = 1u << (i % 32);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, m) != hipSuccess) { printf("bit %d refused\n", i); continue; }
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, s, d);
        hipStreamSynchronize(s);
        unsigned h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        xccs[i] = h[0] & 0xf; cus[i] = (h[1] >> 8) & 0xfff;     // CU id [11:8], SH [12], SE [15:13]
        hipStreamDestroy(s);
    }
    for (int w = 0; w < 8; w++) {
        printf("word %d xcc:", w);
        for (int b = 0; b < 32; b++) printf(" %d", xccs[32 * w + b]);
        printf("\n");
    }
    printf("bits 192..223 cu ids (hex):");
    for (int i = 192; i < 224; i++) printf(" %x", cus[i]);
    printf("\n");
    return 0;
}
