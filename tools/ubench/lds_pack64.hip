// Microbenchmark (round 5): the scatter of one backward sample as 32 x ds_add_u32 (product: one 32-bit
// fixed-point word per slab float, channel-planar) against 16 x ds_add_u64 (two channels packed per 64-bit word), at the
// lane activity and address pattern the kernel has (tools/exp4_stats.py: ~23 active lanes per 32, random cells of an 8^3
// slab with the z-padded layout, 8 corners per sample).  Prints LDS cycles per SAMPLE per CU.
// Build: hipcc --offload-arch=gfx950 -O3 lds_pack64.hip -o lds_pack64 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kVp = 8 * (64 + 5);  // padded voxels per channel plane (bwd_prim_body: gD = TH*TW + 5)

template <int MODE>  // 0: 32 x u32 (4 planes), 1: 16 x u64 (2 planes of channel pairs), 2: 16 x u32 (lower bound: half the atomics)
__global__ __launch_bounds__(128) void k(const int *cell, const int *act, int iters, float *out) {
    __shared__ unsigned long long s64[2 * kVp + 64];
    unsigned *s32 = reinterpret_cast<unsigned *>(s64);
    for (int i = threadIdx.x; i < 2 * kVp + 64; i += 128) s64[i] = 0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const bool on = act[lane] != 0;
    int c0 = cell[lane];
    for (int it = 0; it < iters; ++it) {
        c0 = (c0 * 37 + 11 + it) & 511;  // next random cell
        const int x = c0 & 7, y = (c0 >> 3) & 7, z = c0 >> 6;
        const int gv = (z > 6 ? 6 : z) * (64 + 5) + (y > 6 ? 6 : y) * 8 + (x > 6 ? 6 : x);
        if (on) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int off = gv + (c & 1) + ((c >> 1) & 1) * 8 + (c >> 2) * (64 + 5);
                if (MODE == 0) {
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) atomicAdd(s32 + ch * kVp + off, 1u + ch);
                } else if (MODE == 1) {
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr) atomicAdd(s64 + pr * kVp + off, (1ull << 32) + 1ull + pr);
                } else {
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) atomicAdd(s32 + ch * kVp + off, 1u + ch);
                }
            }
        }
    }
    __syncthreads();
    unsigned long long t = 0;
    for (int i = threadIdx.x; i < 2 * kVp; i += 128) t += s64[i];
    if (t == 0x1234567ull) out[0] = (float)t;
}

template <int MODE>
float run(const int *dc, const int *da, int iters, float *dout) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256 * 6, 128>>>(dc, da, 10, dout);
    hipEventRecord(a);
    k<MODE><<<256 * 6, 128>>>(dc, da, iters, dout);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    int *dc, *da; float *dout;
    hipMalloc(&dc, 256); hipMalloc(&da, 256); hipMalloc(&dout, 4);
    const int iters = 400;
    for (int nact : {64, 46, 32, 23}) {
        std::vector<int> hc(64), ha(64);
        unsigned h = 12345u;
        for (int l = 0; l < 64; ++l) {
            h = h * 1664525u + 1013904223u;
            hc[l] = (h >> 9) & 511;
            ha[l] = ((l * nact) / 64 != ((l + 1) * nact) / 64) ? 1 : 0;  // nact lanes spread evenly over the wave
        }
        hipMemcpy(dc, hc.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(da, ha.data(), 256, hipMemcpyHostToDevice);
        const float t0 = run<0>(dc, da, iters, dout), t1 = run<1>(dc, da, iters, dout), t2 = run<2>(dc, da, iters, dout);
        // wave-samples per CU: 6 blocks/CU * 2 waves * iters; cycles at ~2.4 GHz
        const double ws = 6.0 * 2 * iters;
        auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / ws; };
        printf("%2d active lanes: 32 x ds_add_u32 %.0f   16 x ds_add_u64 %.0f   (16 x ds_add_u32 %.0f)   LDS cycles per wave-sample per CU;"
               "  u64 / u32 per instruction = %.2f\n", nact, cyc(t0), cyc(t1), cyc(t2), 2.0 * cyc(t1) / cyc(t0));
    }
    return 0;
}
