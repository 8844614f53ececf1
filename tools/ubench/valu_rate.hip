// Microbenchmark: issue cost of packed fp32 VALU (v_pk_fma_f32, v_pk_mul_f32) against plain v_fma_f32 / v_mul_f32 / v_mov_b32 /
// v_cvt_rpi_i32_f32 on gfx950, per SIMD, at 1, 2 and 3 waves per SIMD (independent accumulator chains: throughput, not
// latency).  The backward march's sample body is ~250 VALU instructions, a third of them packed; whether a packed op costs
// one issue slot or two decides whether the register-pair moves it needs are worth it.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, float seed, float *out) {
    float a[8];
    v2f p[8];
    for (int j = 0; j < 8; ++j) a[j] = seed + j + threadIdx.x, p[j] = v2f{seed + j, seed - j + (float)threadIdx.x};
    const float m = 1.0000001f, c = 1e-9f;
    const v2f m2 = {m, m}, c2 = {c, c};
    int ia[8];
    for (int j = 0; j < 8; ++j) ia[j] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(m), "v"(c));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(m2), "v"(c2));
                if (MODE == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(m));
                if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(m2));
                if (MODE == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(a[j]) : "v"(a[(j + 1) & 7]));
                if (MODE == 5) asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(ia[j]) : "v"(a[j]));
                if (MODE == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j]) : "v"(c2));
            }
        }
    }
    float t = 0.f;
    for (int j = 0; j < 8; ++j) t += a[j] + p[j].x + p[j].y + (float)ia[j];
    if (t == 12345.f) out[0] = t;
}

template <int MODE>
double run(int blocks_per_cu, float *dout) {
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256 * blocks_per_cu, 256>>>(10, 1.f, dout);
    hipEventRecord(a);
    k<MODE><<<256 * blocks_per_cu, 256>>>(iters, 1.f, dout);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // wave-instructions per SIMD = blocks_per_cu (one wave of each block per SIMD) * iters * 64; cycles at 2.4 GHz
    return ms * 1e-3 * 2.4e9 / ((double)blocks_per_cu * iters * 64);
}

int main() {
    float *dout; hipMalloc(&dout, 4);
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_mul_f32", "v_pk_mul_f32", "v_mov_b32", "v_cvt_rpi_i32_f32", "v_pk_add_f32"};
    for (int w = 1; w <= 3; ++w) {
        const double c[7] = {run<0>(w, dout), run<1>(w, dout), run<2>(w, dout), run<3>(w, dout), run<4>(w, dout), run<5>(w, dout), run<6>(w, dout)};
        printf("%d wave(s) per SIMD, cycles per wave-instruction per SIMD (2.4 GHz nominal):", w);
        for (int i = 0; i < 7; ++i) printf("  %s %.2f", names[i], c[i]);
        printf("\n");
    }
    return 0;
}
