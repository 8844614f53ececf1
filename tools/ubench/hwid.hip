#include <hip/hip_runtime.h>
__global__ void k(unsigned *out) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
}
int main() {
    unsigned *d; hipMalloc(&d, 8 * 4096); k<<<4096, 64>>>(d); unsigned h[8192]; hipMemcpy(h, d, 8 * 4096, hipMemcpyDeviceToHost);
    // print distinct (xcc, se, sh, cu) and the first 40 blocks
    for (int b = 0; b < 48; ++b) printf("b%d xcc %u se %u sh %u cu %u simd %u wave %u\n", b, h[2*b+1] & 0xf, (h[2*b] >> 13) & 7, (h[2*b] >> 12) & 1, (h[2*b] >> 8) & 0xf, (h[2*b] >> 4) & 3, h[2*b] & 0xf);
    return 0;
}
