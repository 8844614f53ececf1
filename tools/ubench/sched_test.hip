// standalone check of the CU-local packet scheduler (experiment): every packet claimed exactly once, no hang
#define MVP_SCHED_SPIN_CAP 2000000
#include "march_packet.h"
#include <cstdio>
#include <vector>
using namespace mvp;
__global__ __launch_bounds__(64) void k(MarchParams p, uint32_t *claimed, uint32_t *err) {
    int n, tidx;
    if (!claim_packet(p, n, tidx)) return;
    if (lane_id() == 0) {
        if (n < 0 || n >= p.N || tidx < 0 || tidx >= p.tiles_x * p.tiles_y) atomicAdd(err, 1u);
        else atomicAdd(claimed + (size_t)n * p.tiles_x * p.tiles_y + tidx, 1u);
    }
    // pretend to work for a data-dependent while
    float a = 1.f;
    for (int i = 0; i < 200 + (tidx % 37) * 40; ++i) a = a * 1.0001f + 0.1f;
    if (a == 123.f) err[1] = 1;
}
int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 80, H = argc > 2 ? atoi(argv[2]) : 512, W = argc > 3 ? atoi(argv[3]) : 512;
    MarchParams p = {};
    p.N = N, p.H = H, p.W = W, p.K = 16;
    p.tiles_x = (W + 7) / 8, p.tiles_y = (H + 7) / 8;
    const size_t np = (size_t)N * p.tiles_x * p.tiles_y;
    uint32_t *claimed, *err; unsigned long long *sched;
    hipMalloc(&claimed, np * 4); hipMalloc(&err, 64); hipMalloc(&sched, 8 * 8 * kSchedSlots + 64);
    hipMemset(claimed, 0, np * 4); hipMemset(err, 0, 64); hipMemset(sched, 0, 8 * 8 * kSchedSlots + 64);
    p.sched = sched; p.diag = err + 4;
    const size_t grid = np + 1000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b); hipEventRecord(a);
    k<<<dim3((unsigned)grid), dim3(64)>>>(p, claimed, err);
    hipEventRecord(b); hipError_t e = hipEventSynchronize(b); float ms = 0; hipEventElapsedTime(&ms, a, b);
    std::vector<uint32_t> h(np); uint32_t he[16];
    hipMemcpy(h.data(), claimed, np * 4, hipMemcpyDeviceToHost); hipMemcpy(he, err, 64, hipMemcpyDeviceToHost);
    size_t zero = 0, multi = 0;
    for (size_t i = 0; i < np; ++i) { zero += h[i] == 0; multi += h[i] > 1; }
    printf("N=%d %dx%d packets %zu: unclaimed %zu, claimed twice %zu, bad %u, spin-cap hits [wait %u, leftover %u], %.3f ms, hip %d\n", N, H, W, np, zero, multi, he[0], he[4], he[5], ms, (int)e);
    return 0;
}
