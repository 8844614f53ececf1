// tools/ubench/scalar_store.hip -- do scalar stores (s_store_dwordx2 + s_dcache_wb) work on gfx950, and what do they cost?
// Kernel A: every wave writes 128 ballots through the scalar data cache; kernel B reads them back with scalar loads and checks.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/scalar_store.hip -o tools/ubench/scalar_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void write_masks(unsigned long long *out, const float *x, int nmask) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned long long *base = out + (size_t)wave * nmask;
    base = (unsigned long long *)__builtin_amdgcn_readfirstlane((int)((size_t)base & 0xffffffffu)) == nullptr ? base : base;  // (no-op)
    unsigned long long bs = (unsigned long long)(size_t)base;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)bs), hi = __builtin_amdgcn_readfirstlane((unsigned)(bs >> 32));
    unsigned long long sb = ((unsigned long long)hi << 32) | lo;
    for (int m = 0; m < nmask; ++m) {
        const float v = x[(size_t)wave * 64 + lane] - 0.01f * (float)m * (float)((lane * 7 + m) % 13);
        const unsigned long long mask = __ballot(v > 0.f);
        const unsigned off = (unsigned)m * 8u;
        asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(mask), "s"(sb), "s"(off) : "memory");
    }
    asm volatile("s_dcache_wb" ::: "memory");
}
__global__ void check_masks(const unsigned long long *in, const float *x, int nmask, unsigned *bad) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    for (int m = 0; m < nmask; ++m) {
        const float v = x[(size_t)wave * 64 + lane] - 0.01f * (float)m * (float)((lane * 7 + m) % 13);
        const unsigned long long mask = in[(size_t)wave * nmask + m];
        if ((((mask >> lane) & 1ull) != 0ull) != (v > 0.f)) atomicAdd(bad, 1u);
    }
}
int main() {
    const int waves = 256 * 8 * 4, nmask = 128;
    std::vector<float> hx((size_t)waves * 64);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.3f;
    float *x; unsigned long long *o; unsigned *bad;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&o, (size_t)waves * nmask * 8); hipMalloc(&bad, 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 4); hipMemset(o, 0xff, (size_t)waves * nmask * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL(write_masks, dim3(waves / 8), dim3(512), 0, 0, o, x, nmask);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("write_masks: %d waves x %d scalar stores: %.3f ms (%.1f ns per store per wave)\n", waves, nmask, ms, ms * 1e6 / nmask / (waves / 256.0 / 8.0));
    }
    hipLaunchKernelGGL(check_masks, dim3(waves / 8), dim3(512), 0, 0, o, x, nmask, bad);
    unsigned hb = 123; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("mismatching bits: %u (%s)\n", hb, hb == 0 ? "scalar stores WORK" : "BROKEN");
    return hb != 0;
}
