// Microbenchmark (round 5): the eight-corner gather of one forward sample in the fp32 slab layout
// (8 x global_load_dwordx4, 16 B per voxel, 8 KB per 8^3 slab) against the fp16 layout (8 B per voxel, 4 KB per slab)
// read as (a) 8 x global_load_dwordx2 and (b) 4 x global_load_dwordx4 -- the two x-neighbours of a corner pair are 16
// contiguous bytes, 8-byte aligned.  Lanes pick random cells of random slabs out of a pool whose size sets the L1 / L2
// residency.  Prints texture-path cycles per wave-SAMPLE per CU.
// Build: hipcc --offload-arch=gfx950 -O3 gather_layout.hip -o gather_layout ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(8)));

template <int MODE>  // 0: fp32 layout 8 x 16 B; 1: fp16 layout 8 x 8 B; 2: fp16 layout 4 x 16 B (x pairs)
__global__ __launch_bounds__(256) void k(const char *__restrict__ data, int nslabs, int iters, float *out) {
    unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        const unsigned slab = (h >> 12) % (unsigned)nslabs, c = (h >> 3) & 511u;
        const unsigned x = (c & 7) > 6 ? 6 : (c & 7), y = ((c >> 3) & 7) > 6 ? 6 : ((c >> 3) & 7), z = (c >> 6) > 6 ? 6 : (c >> 6);
        const unsigned vox = z * 64 + y * 8 + x;
        if (MODE == 0) {
            const char *p = data + (size_t)slab * 8192 + vox * 16;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f4 v = *reinterpret_cast<const f4 *>(p + (j & 1) * 16 + ((j >> 1) & 1) * 128 + (j >> 2) * 1024);
                acc += v.x + v.y + v.z + v.w;
            }
        } else if (MODE == 1) {
            const char *p = data + (size_t)slab * 4096 + vox * 8;
            // (inline assembly: written as C++ the backend merges the x-neighbour pairs into dwordx4 loads, which is mode 2)
            f2 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile("global_load_dwordx2 %0, %8, off\n\tglobal_load_dwordx2 %1, %8, off offset:8\n\t"
                         "global_load_dwordx2 %2, %8, off offset:64\n\tglobal_load_dwordx2 %3, %8, off offset:72\n\t"
                         "global_load_dwordx2 %4, %8, off offset:512\n\tglobal_load_dwordx2 %5, %8, off offset:520\n\t"
                         "global_load_dwordx2 %6, %8, off offset:576\n\tglobal_load_dwordx2 %7, %8, off offset:584\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                         : "v"(p)
                         : "memory");
            acc += (v0.x + v0.y) + (v1.x + v1.y) + (v2.x + v2.y) + (v3.x + v3.y) + (v4.x + v4.y) + (v5.x + v5.y) + (v6.x + v6.y) +
                   (v7.x + v7.y);
        } else {
            const char *p = data + (size_t)slab * 4096 + vox * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // 16 bytes at an 8-byte aligned address: one dwordx4 gather (never crosses a 64-byte slab row)
                const f4u v = *reinterpret_cast<const f4u *>(p + (j & 1) * 64 + (j >> 1) * 512);
                acc += v.x + v.y + v.z + v.w;
            }
        }
    }
    if (acc == -1.f) out[0] = acc;
}

template <int MODE>
float run(const char *d, int nslabs, int iters, float *o) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256 * 8, 256>>>(d, nslabs, 10, o);
    hipEventRecord(a);
    k<MODE><<<256 * 8, 256>>>(d, nslabs, iters, o);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    const int maxslabs = 65536;
    char *d; float *o;
    hipMalloc(&d, (size_t)maxslabs * 8192 + 64); hipMalloc(&o, 4);
    hipMemset(d, 0, (size_t)maxslabs * 8192 + 64);
    const int iters = 100;
    for (int ns : {2, 64, 4096, 65536}) {  // 16 KB (L1-resident in fp32), 512 KB (L2), 32 MB (fp32: past the 4 MB L2), 512 MB (HBM)
        const float t0 = run<0>(d, ns, iters, o), t1 = run<1>(d, ns, iters, o), t2 = run<2>(d, ns, iters, o);
        const double ws = 8.0 * 4 * iters;  // wave-samples per CU
        auto cyc = [&](float ms) { return ms * 1e-3 * 2.4e9 / ws; };
        printf("%6d slabs: fp32 8 x dwordx4 %.0f   fp16 8 x dwordx2 %.0f   fp16 4 x dwordx4 (x pairs) %.0f   cycles per wave-sample per CU\n",
               ns, cyc(t0), cyc(t1), cyc(t2));
    }
    return 0;
}
