// Sustained v_mfma_f32_32x32x16_bf16 rate for the accumulator pattern of csrc/bgmlp.hip: 8 independent 32x32 tiles per
// wave (2 B fragments x 4 A fragments), issued back to back, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k(int iters, float *out) {
    bf16x8 a[2], b[4];
    for (int i = 0; i < 8; ++i) {
        a[0][i] = (__bf16)(float)(threadIdx.x & 3), a[1][i] = (__bf16)1.f;
        for (int j = 0; j < 4; ++j) b[j][i] = (__bf16)(float)(j + (threadIdx.x & 1));
    }
    f32x16 acc[2][4];
    for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 4; ++ni) for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
    }
    float s = 0.f;
    for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 4; ++ni) for (int r = 0; r < 16; ++r) s += acc[mi][ni][r];
    if (s == 12345.f) out[0] = s;
}
template <int THREADS>
void run(int wgs, float *out) {
    const int iters = 4000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<THREADS><<<wgs, THREADS>>>(100, out); hipDeviceSynchronize();
    hipEventRecord(a);
    k<THREADS><<<wgs, THREADS>>>(iters, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double flop = (double)wgs * (THREADS / 64) * iters * 32 * 32768.0;
    printf("%4d workgroups x %d waves: %.3f ms  %.0f TFLOP/s\n", wgs, THREADS / 64, ms, flop / ms * 1e-9);
}
int main() {
    float *out; hipMalloc(&out, 4);
    run<256>(256, out);    // 1 wave / SIMD
    run<512>(256, out);    // 2 waves / SIMD
    run<256>(512, out);    // 2 waves / SIMD as two workgroups
    run<1024>(256, out);   // 4 waves / SIMD
    return 0;
}
