// How many workgroups of a given dynamic-LDS size does a gfx950 CU run at once?  512 workgroups of 256 threads that each
// spin ~100 us: one round (2 per CU) or two rounds (1 per CU) shows in the wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void spin(long long cycles, int *sink) {
    extern __shared__ int smem[];
    smem[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (smem[(threadIdx.x + 1) & 255] == -1) *sink = 1;
}
int main() {
    int *sink; hipMalloc(&sink, 4);
    const int sizes[] = {65536, 80 * 1024, 81920, 81920 - 1280, 82944, 163840 / 2, 163840};
    for (int lds : sizes) {
        hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        spin<<<512, 256, lds>>>(1000, sink); hipDeviceSynchronize();
        hipEventRecord(a);
        spin<<<512, 256, lds>>>(10000, sink);   // wall_clock64 ticks at 100 MHz: 10000 = 100 us
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("dynamic LDS %6d B: 512 workgroups x 100 us -> %.3f ms (%s)\n", lds, ms, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
