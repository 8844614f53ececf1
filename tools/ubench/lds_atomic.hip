// Microbenchmark: cost of LDS float/integer atomics on gfx950 under different lane-activity and address patterns.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic.hip -o lds_atomic ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>  // 0: ds_add_f32, 1: ds_add_u32, 2: plain read+add+write, 3: ds_add_u64
__global__ __launch_bounds__(256) void k(const int *addr, int active_mod, int iters, float *out) {
    __shared__ float s[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) s[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int a = addr[lane];
    const bool act = (lane % active_mod) == 0;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int ad = (a + j * 67 + it) & 4095;
            if (act) {
                if (MODE == 0) atomicAdd(&s[ad], 1.0f);
                if (MODE == 1) atomicAdd(reinterpret_cast<unsigned *>(s) + ad, 1u);
                if (MODE == 2) s[ad] += 1.0f;
                if (MODE == 3) atomicAdd(reinterpret_cast<unsigned long long *>(s) + (ad >> 1), 1ull);
            }
        }
    }
    __syncthreads();
    float t = acc;
    for (int i = threadIdx.x; i < 4096; i += 256) t += s[i];
    if (t == -1.f) out[0] = t;
}

template <int MODE>
float run(const int *daddr, int active_mod, int iters, float *dout) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256 * 4, 256>>>(daddr, active_mod, 10, dout);
    hipEventRecord(a);
    k<MODE><<<256 * 4, 256>>>(daddr, active_mod, iters, dout);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    int *daddr; float *dout;
    hipMalloc(&daddr, 64 * 4); hipMalloc(&dout, 4);
    const int iters = 200;
    const char *pat[] = {"distinct consecutive", "all same", "groups of 4 same", "stride 4 (AoS ch)", "random"};
    for (int p = 0; p < 5; ++p) {
        std::vector<int> h(64);
        for (int l = 0; l < 64; ++l) {
            if (p == 0) h[l] = l;
            if (p == 1) h[l] = 7;
            if (p == 2) h[l] = l / 4;
            if (p == 3) h[l] = l * 4;
            if (p == 4) h[l] = (l * 2654435761u >> 7) & 4095;
        }
        hipMemcpy(daddr, h.data(), 256, hipMemcpyHostToDevice);
        for (int am : {1, 2, 4, 16}) {
            float t0 = run<0>(daddr, am, iters, dout), t1 = run<1>(daddr, am, iters, dout), t2 = run<2>(daddr, am, iters, dout),
                  t3 = run<3>(daddr, am, iters, dout);
            // wave-instructions per CU: 4 blocks/CU * 4 waves * iters * 32; cycles at ~2.4 GHz
            const double winst = 4.0 * 4 * iters * 32;
            auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / winst; };
            printf("%-22s active 1/%-2d : f32 %.1f  u32 %.1f  rmw %.1f  u64 %.1f  LDS cycles/wave-instr (per CU)\n", pat[p], am,
                   cyc(t0), cyc(t1), cyc(t2), cyc(t3));
        }
    }
    return 0;
}
