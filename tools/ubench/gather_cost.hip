// Microbenchmark: vector-memory gather cost (global_load_dwordx4, L1/L2-resident data) vs. number/arrangement of
// active lanes on gfx950.  Measured on MI355X, cycles per wave instruction per CU:
//   lanes              64     32     16      8      4      1
//   L1-missing (512K)  143    71     35     20.6   20.9   19.2     (2.24 cycles per lane, floor 20)
//   L1-resident (16K)  41     22-29  18-23  20.0   20.9   19.7     (a full wave costs 2x the floor)  Build: hipcc --offload-arch=gfx950 -O3 gather_cost.hip -o gather_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void k(const float4 *__restrict__ data, int nvox, unsigned long long mask, int iters,
                                         float *out) {
    const int lane = threadIdx.x & 63;
    const bool act = (mask >> lane) & 1ull;
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    float4 acc = make_float4(0, 0, 0, 0);
    if (act) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                h = h * 1664525u + 1013904223u;
                const float4 v = data[(h >> 8) % (unsigned)nvox];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == -1.f) out[0] = acc.x;
}

int main(int argc, char **argv) {
    // argv[1] = number of voxels: default 512*64 (64 slabs of 8 KB = 512 KB: L2-resident, L1-missing); 1024 (16 KB) = L1-resident
    const int nvox = argc > 1 ? atoi(argv[1]) : 512 * 64;
    float4 *d; float *o;
    hipMalloc(&d, nvox * 16); hipMalloc(&o, 4);
    hipMemset(d, 0, nvox * 16);
    struct { const char *name; unsigned long long m; } pats[] = {
        {"64 lanes", ~0ull}, {"32 lanes (low half)", 0xffffffffull}, {"32 lanes (every other)", 0x5555555555555555ull},
        {"16 lanes (low quarter)", 0xffffull}, {"16 lanes (every 4th)", 0x1111111111111111ull},
        {"8 lanes (every 8th)", 0x0101010101010101ull}, {"4 lanes (one quad)", 0xfull}, {"1 lane", 1ull}};
    const int iters = 200;
    for (auto &p : pats) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        k<<<256 * 8, 256>>>(d, nvox, p.m, 10, o);
        hipEventRecord(a);
        k<<<256 * 8, 256>>>(d, nvox, p.m, iters, o);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double winst = 8.0 * 4 * iters * 8;  // wave-instructions per CU
        printf("%-26s %.1f cycles per gather wave-instruction per CU (at 2.4 GHz)\n", p.name, ms * 1e-3 * 2.4e9 / winst);
    }
    return 0;
}
