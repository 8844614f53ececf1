// march_bwd.hip -- the backward march: primitive-centric kernel (bwd_prim_kernel), its two-pass instantiation, the
// per-packet bound prologue, and the C-ABI entry point mvp_march_backward, which also launches the ray-centric
// kernel of march_packet.h behind them for whatever they handed over.
//   /root/reference/extensions/mvpraymarch/mvpraymarch.cpp:68-100 (raymarch_backward_cuda),
//   mvpraymarch_subset_kernel.h:102-216, primaccum.h:81-98, primsampler.h:68-91, utils.h:504-643, primtransf.h:155-179
#include "march_packet.h"

namespace mvp {

// =================================================================================================
// Primitive-centric backward.  One workgroup (4 waves) per (image n, primitive k).
//   LDS: [V] float4 template slab | [2][Vp] int64 fixed-point gradient, two channels per word: (r | g), (b | a) |
//        ray queue (kQueueCap x 8 B) | small reduce area.   Vp = padded voxel count (z stride TH*TW + kGradPadZ, see below;
//        the warp-field variant has no pad: a cell index serves the slab read and its scatter) | warp-field variant only:
//        [VW] float4 warp grid | [VW] int64 (x | y) + [VW] int32 z fixed-point sums of grad_warp.
//   Work proceeds in rounds of 8 list entries (ray packets):
//     phase 1 (lanes = the rays the round's list records NAME -- each record carries the forward's mask of the packet's rays
//             that have a lattice step in this box, ~40 % of them on head-like scenes; a wave compacts the rays its records
//             name and examines them 64 at a time, round 6): exact ray/box interval, clipped to the ray's first step and to
//             the sample that saturated it -> the rays that really have samples here go into the LDS queue, sorted by
//             their number of steps (LDS integer atomics for bucket tickets).
//     phase 2 (lanes = queued rays, evenly split over the 4 waves): every lane walks ITS OWN lattice steps
//             through the box (aligned by entry step), samples the LDS slab, scatters into the LDS gradient.
// Per-sample math: primaccum.h:81-98 with the prefix replaced by the forward's record
//   key <  satkey : weight = alpha*dt, dL_alpha = dt * dot((rgb,1) - (raysat,1 | 0), dL)
//   key == satkey : weight = 1 - alpha_before, dL_alpha = 0          (the sample that saturated the ray)
//   key >  satkey : not evaluated by the forward
// then primsampler.h:68-91, utils.h:504-643 (scatter into the LDS slab), primtransf.h:155-179 (12 sums).
//
// Slab-gradient accumulation.  Measured on MI355X (tools/ubench/lds_atomic.hip): ds_add_f32 retires ~3 cycles
// per ACTIVE LANE (193 cycles per wave64 instruction, any address pattern) while ds_add_u32 takes 4.8 cycles per
// wave instruction when conflict-free.  The 32 contributions of a sample are therefore accumulated in FIXED POINT with
// integer LDS atomics, ONE int32 word per slab float (round 3; rounds 1-2 used a hi/lo pair of words = 64 atomics per
// sample, and the LDS pipe was busy 82 % of the kernel) -- and since round 5 TWO such words per 64-bit atomic (16 ds_add_u64
// per sample: word = hi * 2^32 + lo with signed lo, acc_read4; measured at the kernel's lane activity a ds_add_u64 costs 1.45 x
// a ds_add_u32, tools/ubench/lds_pack64.hip; C2 5.76 -> 5.63 ms, C4 0.98 -> 0.94):  acc += rn(value * s),  s = 0.999 * 2^31 / (n * B)  where
//   * n is the EXACT number of samples of the current round (counted while the rays are queued) and B bounds any single
//     contribution of the round, so |sum| < 2^31: no overflow.  B = G_q * min(1, Amax * dt) for the colour channels:
//     G_q = max |grad_rayrgba| over the ray PACKETS of the round's list entries (packetmax_kernel, one pass over the
//     upstream gradient before this kernel), Amax = the slab's max |opacity|; a plain sample weighs
//     alpha * fade * dt <= Amax * dt, and the sample that saturates a ray weighs 1 - alpha_before <= its own alpha * dt (it
//     saturated BECAUSE alpha_before + alpha * dt >= 1).  Opacity channel: B = dt * (3 (Tmax + Rmax) + 1) * G_q.
//     Resolution 2^-31 * n * B: a round of typical packets (C2: ~1200 samples) resolves 2^-20.8 of the bound; rounds are
//     cut so that n <= 2^14 (a round takes fewer list entries when the packets' step ranges are long).
//   * DYNAMIC RANGE.  One word resolves the round's contributions relative to the LARGEST upstream gradient near it.
//     While it marches, the round records the smallest max |g| of the rays it really marched (rays whose upstream gradient
//     is exactly zero contribute exact zeros and are skipped); when that is more than 256x below G_q -- an outlier pixel
//     in one of the packets, whether its ray crosses the box or not -- the primitive is left to the TWO-PASS instantiation
//     of this kernel (RESID; a small persistent grid launched right behind, which returns at once when no primitive was
//     marked): every round is marched twice there, pass A's sums of rn(x) are flushed, pass B accumulates the residuals
//     rn((x - rn(x)) * 2^31 / n) -- together 2^-62 * n^2 * B, finer than fp32.  Never on uniform or Gaussian upstream
//     gradients (L1 / L2 image losses: P ~ 4e-8 per ray); on heavy-tailed ones it keeps every primitive exact where it
//     matters.  (The residual scatter lives in its own instantiation because its mere presence in this kernel -- 21
//     spilled VGPRs in a branch never taken -- cost 5 % at C2 and 15 % at C3.)
//   * a primitive whose list needs more than one round flushes the sums at the end of every round but the last into
//     grad_template itself (every voxel has one owner thread, at every flush and at the end, so no atomics and no second
//     LDS array; each flush converts with its round's scale) and restarts from zero.
//   * ACCUMULATED ROUNDING.  Every add rounds to a quantum q_r = n_r B_r / 2^31, a cell receives ~8 n_r / V of them per
//     round, so after the rounds r a cell's sum carries noise of about 0.29 / 2^31 * sqrt(8 / V * sum_r n_r^3) times the
//     bound.  Ordinary primitives (C2: ~1200 samples, V = 512) sit at 1e-7 of the bound; a box that fills the image
//     (tens of thousands of samples over several full rounds) reaches 2e-4 -- per-mille errors of ITS OWN gradient when
//     the values are far below the bound.  The sum over rounds of n_r^3 is therefore held below kNoiseBudget * V (noise
//     3e-6 of the bound, rms): every round may spend the share its entries have in the list, and a round over its share is
//     CUT -- fewer entries per round from there on (round 6; before, such a primitive went to the two-pass instantiation,
//     which only a single record over its share still does).
// The sums are exact integers, so a round's result does not depend on the order its samples arrive in.  The forward
// appends list entries in a different order on every run; a multi-round primitive therefore walks its entries in
// ascending packet order (a rank sort of the keys at kernel start, indices in LDS), which makes the composition of every
// round -- hence every scale, every pass decision and every flushed float -- the same on every run: the slab gradient is
// BIT-REPRODUCIBLE run to run for every primitive (the fp32-atomic formulation is not; the two-word form was, up to
// 65536 samples per primitive).
// A sample whose weight breaks the bound (signed opacity) is detected and the primitive is handed to the ray-centric
// kernel, like one with a non-finite bound.
// The gradient arrays use a z stride of TH*TW + kGradPadZ words.  With the natural stride (a multiple of the 32 banks)
// two layers of cells collide bank for bank.  Measured at C2 (tools/exp4_stats.py): a 32-lane group has ~23 active
// lanes, at most ~2.05 of them on one address, and the busiest bank serves 3.25 lanes with pad 4 but 2.93 with pad 5
// (other (y stride, z stride) pairs tried: 2.90-3.20; the lanes' cells are close to random, so ~2.9 is the floor for
// a linear layout).
// =================================================================================================
// Waves per workgroup (one workgroup = one primitive) is a template parameter of the kernel, PW in {2, 3}.  The kernel is
// latency-bound per workgroup and holds ~164 VGPRs (12 waves per CU): 3 waves x 4 workgroups per CU keeps one more
// primitive in flight than round 1's 4 x 3, 2 waves x 5 workgroups two more.  Which is faster depends on how much work a
// primitive has: K = 16384 at 512^2 (few packets per primitive) prefers 2 waves (C3 backward 0.87 -> 0.74 ms), K = 8192 at
// 1024^2 prefers 3 (C4 1.90 vs 2.01 ms), C2 is indifferent; the host picks by packets per primitive.
constexpr float kFixRange = 0.999f * 2147483648.f;  // |sum of a round's contributions * scale| stays below 2^31
constexpr float kTwoPassRatio = 256.f;  // marched rays' gradient magnitudes further below the bound than this: two passes
constexpr float kNoiseBudget = 6.2e7f;  // sum over rounds of (samples of the round)^3 per voxel: beyond it, two passes (header)
constexpr int kRoundBudgetLog2 = 14;   // a round takes list entries while 64 lanes x their step ranges stay below 2^14
constexpr int kGradPadZ = 5;  // see the note on the gradient arrays above
constexpr int kEntriesPerWave = 5;  // list entries (packets) each wave examines per round
__host__ __device__ constexpr int prim_entries_per_round(int pw) { return pw * kEntriesPerWave; }  // typical lists: ONE round
__host__ __device__ constexpr int prim_queue_cap(int pw) { return prim_entries_per_round(pw) * 64; }  // rays per round
constexpr uint32_t kRotMul = 5u, kRotMask = 7u;  // per-lane start offset of the walk (bwd_prim_body, phase 2)
constexpr int kLenBuckets = 32;     // rays are queued sorted by their number of lattice steps

// Backward prologue.  (1) Per ray packet (8x8 pixels): max |grad_rayrgba| -> pmax[packet] as float bits (non-negative
// floats order like uints; a NaN's pattern is larger than Inf's, so it is sticky).  The primitive-centric kernel derives
// every round's fixed-point scale from the packets of THAT round, so one outlier pixel costs resolution only where it is.
// (2) Undo what an earlier backward over the same forward left in the hand-off buffer (retain_graph / several losses):
// the "handed over" bit of the counters and flag.
__global__ __launch_bounds__(256) void packetmax_kernel(const float4 *__restrict__ g4, int N, int H, int W, int tiles_x,
                                                        int tiles_y, uint32_t *__restrict__ pmax,
                                                        uint32_t *__restrict__ counts, size_t ncounts,
                                                        uint32_t *__restrict__ tail) {
    const int lane = lane_id();
    const size_t nwaves = (size_t)gridDim.x * (blockDim.x / kWave);
    const size_t w0 = (size_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    const size_t T = (size_t)tiles_x * tiles_y;
    for (size_t pk = w0; pk < (size_t)N * T; pk += nwaves) {
        const size_t n = pk / T;
        const int tidx = (int)(pk - n * T);
        const int ty = tidx / tiles_x, tx = tidx - ty * tiles_x;
        const int px = tx * kTile + (lane & 7), py = ty * kTile + (lane >> 3);
        uint32_t mi = 0u;
        if (px < W && py < H) {
            const float4 v = g4[(n * H + py) * W + px];
            mi = max(max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu),
                     max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu));
        }
        mi = (uint32_t)wave_max((int)mi);  // all patterns are < 2^31: signed max is the same order
        if (lane == 0) pmax[pk] = (pmax[pk] & kPacketFwdOverflow) | ((mi + 3u) >> 2);
    }
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncounts; i += stride) {
        const uint32_t c = counts[i];
        if (c & ~kCountMask) counts[i] = c & kCountMask;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t f = tail[0];
        if (f & (kFlagBwdHandoff | kFlagBwdPrecise)) tail[0] = f & ~(kFlagBwdHandoff | kFlagBwdPrecise);
    }
}

// The four fixed-point sums of gradient cell gv.  PACKED (every instantiation since round 6): two 64-bit words, word = hi * 2^32 + lo with SIGNED
// lo -- channels (r | g) in plane 0 and (b | a) in plane 1 -- so that a sample's scatter is 16 ds_add_u64 instead of 32
// ds_add_u32: lo by sign extension, hi = (word - lo) >> 32, exact while |sum lo| < 2^31 (the per-round bound).
template <bool PACKED>
__device__ __forceinline__ void acc_read4(const int *s_acc, int Vp, int gv, int &a, int &b, int &c, int &d) {
    if constexpr (PACKED) {
        const long long *q = reinterpret_cast<const long long *>(s_acc);
        const long long w0 = q[gv], w1 = q[Vp + gv];
        a = (int)w0, b = (int)((w0 - (long long)a) >> 32);
        c = (int)w1, d = (int)((w1 - (long long)c) >> 32);
    } else {
        a = s_acc[gv], b = s_acc[Vp + gv], c = s_acc[2 * Vp + gv], d = s_acc[3 * Vp + gv];
    }
}
template <bool PACKED>
__device__ __forceinline__ void acc_clear4(int *s_acc, int Vp, int gv) {
    if constexpr (PACKED) {
        long long *q = reinterpret_cast<long long *>(s_acc);
        q[gv] = 0ll, q[Vp + gv] = 0ll;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) s_acc[c * Vp + gv] = 0;
    }
}
__device__ __forceinline__ uint32_t abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }

// float -> int, round to nearest (ties up): v_cvt_rpi_i32_f32.  (int)x truncates toward zero, a systematic shrink of
// every contribution by half a unit on average.
__device__ __forceinline__ int fix_rn(float v) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// (WARP, round 6: the zero padding is evaluated without branches, as zero weights on cells the sample reads anyway --
//  march_common.h: tri_zero_pad.)
// TS > 0: the slab is TS^3 (compile-time strides: the 32 atomics and 8 reads of a sample share ONE address register and
// use immediate offsets); TS == 0: any slab size, strides in registers.
// WARP: the warp-field sampler (algo 1, primsampler.h:53-58,82-88): a second LDS slab (the warp grid) and a second set of
// fixed-point accumulators (grad_warp); the template is sampled at warp(y) with zero padding (general strides only).
constexpr int kBwdOcc = 3;  // waves per SIMD the register allocation aims at (4 = at most 128 VGPRs: 20 spilled)
// RESID: the two-pass instantiation (header, DYNAMIC RANGE): owns the primitives the plain one marked, nothing else.
template <bool FADE8, int TS, int PW, bool WARP, bool RESID>
__device__ __forceinline__ void bwd_prim_body(const MarchParams &p, const int block, float4 *smem4) {
    static_assert(!WARP || TS == 0, "the warp-field variant uses run-time slab dimensions");
    constexpr int kPrimWaves = PW, kPrimBlock = PW * 64;
    constexpr int kEntriesPerRound = prim_entries_per_round(PW), kQueueCap = prim_queue_cap(PW);
    const int TD = TS ? TS : p.TD, TH = TS ? TS : p.TH, TW = TS ? TS : p.TW;
    const int V = TD * TH * TW;
    constexpr int kPadZ = WARP ? 0 : kGradPadZ;  // (warp-field variant: cell indices shared with the slab, 8-byte cells)
    const int gH = TW, gD = TH * TW + kPadZ;  // gradient-array strides (words); x stride 1
    const int Vp = TD * gD;
    float4 *s_T = smem4;
    // fixed-point sums: [2][Vp] int64, two channels per word: (r | g), (b | a) (acc_read4 / the scatter of the walk)
    int *s_acc = reinterpret_cast<int *>(smem4 + V);
    uint2 *s_q = reinterpret_cast<uint2 *>(s_acc + 4 * Vp);  // (Vp is even: 8-byte aligned)
    float *s_red = reinterpret_cast<float *>(s_q + kQueueCap);  // 64 floats
    uint32_t *s_qn = reinterpret_cast<uint32_t *>(s_red + 64);
    uint32_t *s_bucket = s_qn + 4;  // kLenBuckets words
    uint16_t *s_perm = reinterpret_cast<uint16_t *>(s_bucket + kLenBuckets);  // pl_cap entries: list index by rank
    uint32_t *s_gext = reinterpret_cast<uint32_t *>(s_red + 62);  // per round: bits(max), bits(min) of the queued rays' max |g|
    // WARP: [warp grid as float4 (x,y,z,-)] [VWp] int64 (x | y) sums, [VWp] int32 z sums -- behind everything else (16-byte
    // aligned: the host sizes the part above as a multiple of 16 bytes)
    const int WD = WARP ? p.WD : 2, WH = WARP ? p.WH : 2, WW = WARP ? p.WW : 2;
    const int VW = WD * WH * WW, gDw = WH * WW + kPadZ, VWp = WD * gDw;
    float4 *s_W = reinterpret_cast<float4 *>(reinterpret_cast<char *>(smem4) + p.prim_lds_base);
    int *s_wacc = reinterpret_cast<int *>(s_W + VW);

    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    const int K = p.K;
    int n, k;
    if (!prim_of_block(p, block, n, k)) return;
    const size_t pk = (size_t)n * K + k;
    uint32_t *tail = p.pl_count + (size_t)p.N * K;  // [0] flags, [1] reserved, [2] bits(Rmax); then per-packet bits(max |g|)
    const uint32_t *pmax_n = tail + 3 + (size_t)n * p.tiles_x * p.tiles_y;

    // Everything this workgroup needs first is requested at once, before any of it is looked at: flags, list length,
    // the primitive's transform (scalar loads: wave-uniform addresses, data no kernel in flight writes) and -- for the
    // 8^3 instantiation -- this thread's two slab voxels, speculatively (an empty list is rare and the read is valid
    // either way).  The former order (counter -> branch -> slab -> barrier -> transform) cost two more dependent
    // global round trips per workgroup.
    const uint32_t flags = cload(tail);
    // (the two-pass instantiation reads what the plain one, an earlier launch on this stream, wrote: a plain load)
    const uint32_t cnt_raw = RESID ? (uint32_t)uni((int)p.pl_count[pk]) : cload(p.pl_count + pk);
    if (RESID && (cnt_raw & (kCountPrecise | kCountDead)) != kCountPrecise) return;  // not marked (or handed over since)
    const uint32_t cnt = cnt_raw & kCountMask;
    const float *qp = p.primpos + pk * 3, *qr = p.primrot + pk * 9, *qs = p.primscale + pk * 3;
    Rec q;  // SGPRs
    q.pos = mk3(cload(qp), cload(qp + 1), cload(qp + 2));
    q.r0 = mk3(cload(qr), cload(qr + 1), cload(qr + 2));
    q.r1 = mk3(cload(qr + 3), cload(qr + 4), cload(qr + 5));
    q.r2 = mk3(cload(qr + 6), cload(qr + 7), cload(qr + 8));
    q.scale = mk3(cload(qs), cload(qs + 1), cload(qs + 2));
    const float4 *T4 = reinterpret_cast<const float4 *>(p.tplate) + pk * (size_t)V;
    constexpr int kVoxPerThread = (512 + kPrimBlock - 1) / kPrimBlock;  // 8^3 slab: voxels staged per thread
    float4 tv[kVoxPerThread];
    if (TS == 8) {
#pragma unroll
        for (int i = 0; i < kVoxPerThread; ++i)
            tv[i] = (tid + i * kPrimBlock < 512) ? MVP_STREAM_LOAD(T4 + tid + i * kPrimBlock) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 *gT4 = reinterpret_cast<float4 *>(p.grad_tplate) + pk * (size_t)V;
    const uint4 *list = p.pl_list + pk * (size_t)p.pl_cap;  // {key, step range, ray mask lo, hi} per entry
    bool dead = (flags & kFlagGlobal) != 0u || cnt > (uint32_t)p.pl_cap;  // the ray-centric kernel owns it
    // ... and marches only the packets somebody asked it to: the ones the forward could not append are marked already,
    // the recorded ones are marked here, by whoever hands a primitive over
    auto want_ray_centric = [&]() {
        uint32_t *pw = tail + 3 + (size_t)n * p.tiles_x * p.tiles_y;
        const uint32_t m = min(cnt, (uint32_t)p.pl_cap);
        for (uint32_t e = tid; e < m; e += kPrimBlock) atomicOr(pw + (list[e].x >> 9), kPacketBwdWanted);
    };
    if (dead && (flags & kFlagGlobal) == 0u && !RESID) want_ray_centric();  // (list overflow; the global flag marches all)

    // ---- stage the slab with its max |rgb| and max |opacity|; longest step range on the list ----
    // (as bit patterns of |x|: non-negative floats order like uints and a NaN's pattern is above Inf's, so ONE non-finite
    //  voxel makes the bound non-finite -- fmaxf would drop a NaN and the integer sums would turn its contributions into zeros)
    uint32_t tmaxb = 0u, amaxb = 0u;
    float tmax = 0.f, amax = 0.f;
    uint32_t maxlen = 1u;  // longest packet step range on the list (a ray's own range is inside its packet's)
    if (!dead && cnt > 0u) {
        for (uint32_t e = tid; e < cnt; e += kPrimBlock) {
            const uint32_t rg = list[e].y;
            // (a ray with more than 127 steps in this box sends the primitive to the ray-centric kernel: phase 1)
            maxlen = max(maxlen, min((rg >> 16) - (rg & 0xffffu) + 1u, 127u));
        }
        if constexpr (WARP) {
            const float *Wg = p.warp + pk * (size_t)VW * 3;
            for (int v = tid; v < VW; v += kPrimBlock) s_W[v] = make_float4(Wg[v * 3], Wg[v * 3 + 1], Wg[v * 3 + 2], 0.f);
            for (int v = tid; v < 3 * VWp; v += kPrimBlock) s_wacc[v] = 0;
        }
        if (TS == 8) {
#pragma unroll
            for (int i = 0; i < kVoxPerThread; ++i) {
                if (tid + i * kPrimBlock < 512) s_T[tid + i * kPrimBlock] = tv[i];
                tmaxb = max(tmaxb, max(max(abs_bits(tv[i].x), abs_bits(tv[i].y)), abs_bits(tv[i].z)));
                amaxb = max(amaxb, abs_bits(tv[i].w));
            }
        } else {
            for (int v = tid; v < V; v += kPrimBlock) {
                const float4 t = T4[v];
                s_T[v] = t;
                tmaxb = max(tmaxb, max(max(abs_bits(t.x), abs_bits(t.y)), abs_bits(t.z)));
                amaxb = max(amaxb, abs_bits(t.w));
            }
        }
        {  // clear the sums: 4 * Vp words from a 16-byte aligned base, 16-byte stores
            float4 *z4 = reinterpret_cast<float4 *>(s_acc);
            const int nz4 = (4 * Vp) >> 2;
            for (int v = tid; v < nz4; v += kPrimBlock) z4[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int v = (nz4 << 2) + tid; v < 4 * Vp; v += kPrimBlock) s_acc[v] = 0;
        }
        tmaxb = (uint32_t)wave_max((int)tmaxb);  // (patterns < 2^31: signed order)
        amaxb = (uint32_t)wave_max((int)amaxb);
        maxlen = (uint32_t)wave_max((int)maxlen);
        if (lane == 0)
            s_red[wave] = __uint_as_float(tmaxb), s_red[8 + wave] = __uint_as_float(amaxb), s_red[12 + wave] = __uint_as_float(maxlen);
    }
    __syncthreads();
    // What the bounds of a round's contributions are made of, besides the round's own max |grad_rayrgba| G_q (header):
    //   colour:  |w_c * dLs.rgb| <= wrgb * G_q,  wrgb = wmax = min(1, Amax * dt) >= every |sample weight| (checked per sample)
    //   opacity: |w_c * dLs.a|   <= fa * G_q,    fa = dt * (3 (Tmax + Rmax) + 1)        (fade <= 1)
    //   WARP: a corner of the warp grid receives w_c * dL/dy1, |w_c| <= 1 and dL/dy1_x = (TW-1)/2 * sum over corners of
    //         +-w_y w_z (value_c . dLs) with sum |w_y w_z| <= 2, |value_c . dLs| <= (3 Tmax wrgb + Amax fa) G_q
    float wmax = 1.f, wrgb = 1.f, fa = 1.f, fw = 1.f;
    if (!dead && cnt > 0u) {
        tmaxb = __float_as_uint(s_red[0]), amaxb = __float_as_uint(s_red[8]), maxlen = __float_as_uint(s_red[12]);
#pragma unroll
        for (int w = 1; w < kPrimWaves; ++w)
            tmaxb = max(tmaxb, __float_as_uint(s_red[w])), amaxb = max(amaxb, __float_as_uint(s_red[8 + w])),
            maxlen = max(maxlen, __float_as_uint(s_red[12 + w]));
        tmax = __uint_as_float(tmaxb), amax = __uint_as_float(amaxb);
        const float Rmax = __uint_as_float(cload(tail + 2));
        wmax = fminf(1.f, amax * p.stepsize * 1.0001f);
        // (a fully transparent slab -- relu(alpha) = 0 everywhere -- has wmax = 0: its rgb contributions are exact zeros and
        //  any scale serves; its opacity gradient is not zero)
        wrgb = fmaxf(wmax, 9.5367431640625e-07f);
        fa = p.stepsize * (3.f * (tmax + Rmax) + 1.f);
        if constexpr (WARP) fw = (float)(max(TD, max(TH, TW)) - 1) * (3.f * tmax * wrgb + amax * fa);
        // non-finite slab (a NaN / Inf voxel in any channel) / raysat: the ray-centric kernel's case
        if (!(fa < 1.0e30f) || !(fw < 1.0e30f) || !(amax < 1.0e30f)) {
            dead = true;
            want_ray_centric();
            if (tid == 0) {
                atomicOr(p.pl_count + pk, kCountDead);
                raise_flag(tail, kFlagBwdHandoff);
            }
        }
    }
    // (workgroup-uniform values computed from LDS reads: moved to SGPRs, the march's VGPR budget has no room for them)
    wmax = uni(wmax), wrgb = uni(wrgb), fa = uni(fa), maxlen = (uint32_t)uni((int)maxlen);
    if constexpr (WARP) fw = uni(fw);
    // list entries per round: all the workgroup can look at (kEntriesPerRound), fewer when the packets' step ranges are long
    // (64 lanes x range x entries <= 2^kRoundBudgetLog2 keeps a round's sample count, hence its scale exponent c, small)
    // (both can change once: a round whose rounding noise would pass its share of the budget is cut -- see the round loop)
    uint32_t epr = min((uint32_t)kEntriesPerRound, max(1u, (1u << kRoundBudgetLog2) / (64u * maxlen)));
    bool multi = cnt > epr;  // more than one round: walk the entries in ascending key order (see the header)
    __syncthreads();  // s_red is reused below
    if (cnt == 0u || dead) {  // this launch doubles as the zero-fill of the gradient buffers
        if constexpr (WARP) {
            float *gW = p.grad_warp + pk * (size_t)VW * 3;
            for (int v = tid; v < VW * 3; v += kPrimBlock) gW[v] = 0.f;
        }
        for (int v = tid; v < V; v += kPrimBlock) gT4[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tid < 9) p.grad_primrot[pk * 9 + tid] = 0.f;
        if (tid < 3) p.grad_primscale[pk * 3 + tid] = 0.f;
        if (tid < 3) p.grad_primpos[pk * 3 + tid] = 0.f;
        return;
    }

    const float dt = p.stepsize;
    // per-image base pointers are wave-uniform (SGPR pairs); rays are addressed with a 32-bit index inside the image,
    // so every per-lane load is "scalar base + 32-bit vector offset" instead of a 64-bit address held in two VGPRs
    const size_t img = (size_t)n * p.H * p.W;
    const float *raypos_n = p.raypos + img * 3, *raydir_n = p.raydir + img * 3, *tminmax_n = p.tminmax + img * 2;
    const float *grad_n = p.grad_rayrgba + img * 4, *raysat_n = p.raysat_in + img * 3;
    const uint32_t *aux_n = p.rayaux + img * 4;
    const int sW = 1, sH = TW, sD = TH * TW;  // voxel strides of the template slab
    const float mx = 0.5f * (float)(TW - 1), my = 0.5f * (float)(TH - 1), mz = 0.5f * (float)(TD - 1);
    const float mwx = 0.5f * (float)(WW - 1), mwy = 0.5f * (float)(WH - 1), mwz = 0.5f * (float)(WD - 1);  // (warp grid)
    const float nfs_log2e = -p.fadescale * 1.44269504088896341f;  // exp(-fadescale * e) = exp2(nfs_log2e * e): one multiply
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    float c00 = 0.f, c01 = 0.f, c02 = 0.f, c10 = 0.f, c11 = 0.f, c12 = 0.f, c20 = 0.f, c21 = 0.f, c22 = 0.f;

    if (tid == 0) s_qn[2] = 0u;
    // rank of every entry among the list's keys ((packet << 9) | slot: one entry per packet, all different); the key stream is
    // wave-uniform -> scalar loads, four entries (64 bytes; pl_cap is a multiple of 4) at a time
    // (ONE call site.  Where a round is cut, below, the ranks are made by a plain loop; an outer "start over" loop around this
    //  call was built too: 4 spilled VGPRs and 10 more spilled SGPRs for the loop-carried state of a restart)
    auto rank_entries = [&]() {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // (a native vector: HIP's uint4 class has no
        typedef const __attribute__((address_space(4))) u32x4 *cu4;   //  constructor from another address space)
        const cu4 l4 = reinterpret_cast<cu4>(reinterpret_cast<uintptr_t>(list));   // one entry per 16-byte element
        for (uint32_t e = tid; e < cnt; e += kPrimBlock) {
            const uint32_t key = list[e].x;
            uint32_t rank = 0u;
            for (uint32_t j = 0; j < cnt; j += 4u) {
                const uint32_t k0 = l4[j].x, k1 = l4[j + 1u].x, k2 = l4[j + 2u].x, k3 = l4[j + 3u].x;
                rank += (k0 < key ? 1u : 0u) + ((j + 1u < cnt && k1 < key) ? 1u : 0u) +
                        ((j + 2u < cnt && k2 < key) ? 1u : 0u) + ((j + 3u < cnt && k3 < key) ? 1u : 0u);
            }
            s_perm[rank] = (uint16_t)e;
        }
    };
    if (multi) rank_entries();
    bool wbad = false;      // some sample weight was outside the bound: the integer sums cannot be trusted
    bool drained = false;   // grad_template holds the flushed sums of earlier rounds / passes (workgroup-uniform)
    float s_rgb = 1.f, s_a = 1.f, s_w = 1.f;  // this round's scales (workgroup-uniform)
    float cur_mul = 1.f;    // ... times this in the pass being marched (1, or the residual multiplier of pass B)
    // More sums follow (another pass, another round): move the integer sums, divided by their scale, into grad_template
    // itself and restart from zero.  Every voxel is owned by one thread, here and at the end, so the partial sums need
    // no atomics and no second LDS array: written by the first flush, added to by later ones.
    auto flush_sums = [&](float i_rgb, float i_a, float i_w) {
        size_t pkd = pk;
        int td = tid;
        asm volatile("; flush addresses are made here" : "+s"(pkd), "+v"(td));
        float4 *gd = reinterpret_cast<float4 *>(p.grad_tplate) + pkd * (size_t)V;
        for (int v = td; v < V; v += kPrimBlock) {
            const int z = v / sD, rem = v - z * sD;
            const int gv = z * gD + rem;
            float4 g;
            int ia_, ib_, ic_, id_;
            acc_read4<true>(s_acc, Vp, gv, ia_, ib_, ic_, id_);
            g.x = (float)ia_ * i_rgb, g.y = (float)ib_ * i_rgb, g.z = (float)ic_ * i_rgb, g.w = (float)id_ * i_a;
            if (drained) {
                const float4 o_ = gd[v];
                g.x += o_.x, g.y += o_.y, g.z += o_.z, g.w += o_.w;
            }
            gd[v] = g;
            acc_clear4<true>(s_acc, Vp, gv);
        }
        if constexpr (WARP) {
            float *gWd = p.grad_warp + pkd * (size_t)VW * 3;
            const int sDw = WH * WW;
            for (int v = td; v < VW; v += kPrimBlock) {
                const int z = v / sDw, rem = v - z * sDw;
                const int gv = z * gDw + rem;
                const long long w0 = reinterpret_cast<const long long *>(s_wacc)[gv];  // (x | y), like acc_read4
                const int ix_ = (int)w0, iy_ = (int)((w0 - (long long)ix_) >> 32), iz_ = s_wacc[2 * VWp + gv];
                float gx_ = (float)ix_ * i_w, gy_ = (float)iy_ * i_w, gz_ = (float)iz_ * i_w;
                if (drained) gx_ += gWd[v * 3], gy_ += gWd[v * 3 + 1], gz_ += gWd[v * 3 + 2];
                gWd[v * 3] = gx_, gWd[v * 3 + 1] = gy_, gWd[v * 3 + 2] = gz_;
                reinterpret_cast<long long *>(s_wacc)[gv] = 0ll, s_wacc[2 * VWp + gv] = 0;
            }
        }
        drained = true;
        __syncthreads();
    };
    bool pass_b = false;  // this iteration re-marches the round it has just marched, for the residuals (workgroup-uniform)
    for (uint32_t ebase = 0; ebase < cnt;) {
        if (tid < kLenBuckets) s_bucket[tid] = 0u;
        if (tid == 0) s_qn[1] = 0u, s_gext[0] = 0u, s_gext[1] = 0x7fffffffu;
        __syncthreads();
        // The transform is block-uniform and lives in SGPRs.  Made opaque once per round, so that packed-math operand
        // pairs built from it are re-made here (a v_mov each) instead of being carried, spilled, across the march.
        asm volatile("; round-local transform"
                     : "+s"(q.pos.x), "+s"(q.pos.y), "+s"(q.pos.z), "+s"(q.r0.x), "+s"(q.r0.y), "+s"(q.r0.z),
                       "+s"(q.r1.x), "+s"(q.r1.y), "+s"(q.r1.z), "+s"(q.r2.x), "+s"(q.r2.y), "+s"(q.r2.z),
                       "+s"(q.scale.x), "+s"(q.scale.y), "+s"(q.scale.z));
        // ---------------- phase 1: which rays of these packets cross the box, and over which steps ----------------
        // Each wave owns up to kEntriesPerWave entries of the round.  An entry names a ray packet and (round 6) carries the mask
        // of the packet's rays that have a lattice step in this box by the FORWARD's exact test -- ~40 % of them on head-like
        // scenes.  The wave first COMPACTS the rays its entries name into one dense sequence (scalar popcount prefix per entry,
        // v_mbcnt inside it; the records go through this wave's part of the ray queue, which is free until the barrier below),
        // then examines them 64 at a time: exact interval with the formulas of the forward, clipped to the packet's range and
        // the ray's first step -- ~2 passes instead of one per entry at C2.  A live ray takes a ticket in the bucket of its step
        // count (LDS integer atomic), buckets are prefix-summed, and the ray is written at its sorted position, so the 64 rays
        // a wave marches together have (nearly) the same number of steps.
        const uint32_t eend = min(cnt, ebase + epr);
        uint2 item[kEntriesPerWave];  // {ray index inside the image | list slot << 23, first step | steps << 16}
        uint32_t ticket[kEntriesPerWave];
        bool live2[kEntriesPerWave];
        bool toolong = false;
        uint32_t mylen = 0u;
        uint32_t gq_hi = 0u;  // bits of the largest max |grad_rayrgba| over the packets this wave's records name (wave-uniform)
        uint32_t ncmp = 0u;   // rays named by this wave's entries (wave-uniform, <= kEntriesPerWave * 64)
        uint2 *s_cmp = s_q + wave * (kEntriesPerWave * 64);
        {
            // the wave's entries (scalar loads, 16 bytes each).  (Round 6 also measured them as ONE batch of unconditional loads
            // with the packets' gradient bounds behind them -- one wait instead of a dependent round trip per entry: C2 +0.5 %,
            // C3 +2 % SLOWER, 14 more spilled SGPRs: profiles/r06_backward_experiments.json)
            uint32_t ekey[kEntriesPerWave], erg[kEntriesPerWave], emlo[kEntriesPerWave], emhi[kEntriesPerWave];
#pragma unroll
            for (int u = 0; u < kEntriesPerWave; ++u) {
                const uint32_t e = ebase + wave + u * (kPrimBlock / kWave);
                ekey[u] = erg[u] = emlo[u] = emhi[u] = 0u;
                if (e < eend) {
                    const uint32_t le = multi ? (uint32_t)uni((int)s_perm[e]) : e;
                    const uint32_t *lw = reinterpret_cast<const uint32_t *>(list + le);  // wave-uniform: scalar loads
                    ekey[u] = cload(lw), erg[u] = cload(lw + 1), emlo[u] = cload(lw + 2), emhi[u] = cload(lw + 3);
                }
            }
            const uint32_t laneoff = (uint32_t)(lane >> 3) * (uint32_t)p.W + (uint32_t)(lane & 7);
#pragma unroll
            for (int u = 0; u < kEntriesPerWave; ++u) {
                const unsigned long long m = ((unsigned long long)emhi[u] << 32) | emlo[u];
                if (m != 0ull) {  // (wave-uniform; an entry past the round's end has an empty mask)
                    const int tidx = (int)(ekey[u] >> 9);
                    const int ty = tidx / p.tiles_x, tx = tidx - ty * p.tiles_x;
                    // (the mask names rays inside the image only: the forward's `active`)
                    const uint32_t rbase = (uint32_t)(ty * kTile) * (uint32_t)p.W + (uint32_t)(tx * kTile);
                    gq_hi = max(gq_hi, (cload(pmax_n + tidx) & kPacketMaxMask) << 2);
                    const uint32_t pos = __builtin_amdgcn_mbcnt_hi(emhi[u], __builtin_amdgcn_mbcnt_lo(emlo[u], ncmp));
                    if ((m >> lane) & 1ull) s_cmp[pos] = make_uint2((rbase + laneoff) | ((ekey[u] & 511u) << 23), erg[u]);
                    ncmp += (uint32_t)__popcll(m);
                }
            }
        }
        // (the records are read back by other lanes of the SAME wave: LDS operations of a wave complete in order, so all it
        //  takes is that the compiler keeps the order too)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int u = 0; u < kEntriesPerWave; ++u) {
            live2[u] = false;
            ticket[u] = 0u;
            item[u] = make_uint2(0u, 0u);
            if ((uint32_t)(u * kWave) < ncmp) {  // (wave-uniform) pass u over the compacted rays
                const uint32_t ci = (uint32_t)(u * kWave + lane);
                const bool have = ci < ncmp;
                const uint2 cr = have ? s_cmp[ci] : make_uint2(0u, 0u);
                const uint32_t r = cr.x & 0x7fffffu, slot = cr.x >> 23;  // ray index inside image n, list slot
                const int elo = (int)(cr.y & 0xffffu), ehi = (int)(cr.y >> 16);
                int slo = 1, shi = 0;
                if (have) {
                    // byte offsets computed in 32 bits: "SGPR base + zero-extended VGPR offset" addressing
                    const f3 o = ld3(at_bytes<float>(raypos_n, r * 12u)), d = ld3(at_bytes<float>(raydir_n, r * 12u));
                    const float2 tt = *at_bytes<float2>(tminmax_n, r * 8u);
                    const uint4 ax = *at_bytes<uint4>(aux_n, r * 16u);  // {key of the saturating sample, -, first step, -}
                    const int incs = (int)ax.z;
                    // the same formulas the forward used for the packet range [elo, ehi] (the union of these over lanes)
                    const f3 r0 = rot_rows(q, o - q.pos) * q.scale, rd = rot_rows(q, d) * q.scale;
                    const f3 ird = mk3(fast_rcp(rd.x), fast_rcp(rd.y), fast_rcp(rd.z));
                    const f3 t0 = mk3((-1.f - r0.x) * ird.x, (-1.f - r0.y) * ird.y, (-1.f - r0.z) * ird.z);
                    const f3 t1 = mk3((1.f - r0.x) * ird.x, (1.f - r0.y) * ird.y, (1.f - r0.z) * ird.z);
                    const float tn = max3f(fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z));
                    const float tf = min3f(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y), fmaxf(t0.z, t1.z));
                    int l0, h0;
                    if (lane_step_range(tn, tf, tt.x, tt.y, dt, l0, h0)) {
                        slo = max(l0, max(elo, incs));
                        // (round 6) ... and nothing behind the sample that saturated the ray was evaluated by the forward
                        // (key = step << 9 | slot, subset_kernel.h:76-97): such steps are not queued at all -- on a trained-like
                        // scene, where 40 % of the rays saturate in the front shell, a fifth of the (ray, primitive) pairs the
                        // lists name lie entirely behind that point and used to ride through the walk as dead lanes.  An
                        // unsaturated ray's key is 0xffffffff: no clip.
                        const int sats = (int)(ax.x >> 9);
                        shi = min(min(h0, ehi), slot <= (ax.x & 511u) ? sats : sats - 1);
                    }
                }
                if (slo <= shi) {
                    // at most 127 steps per queued item (the len field and the buckets assume short crossings); a box
                    // that is deeper than that along some ray is handed to the ray-centric kernel (flagged below)
                    const int len = shi - slo + 1;
                    if (len > 127) toolong = true;
                    live2[u] = true;
                    item[u] = make_uint2(r | (slot << 23), (uint32_t)slo | ((uint32_t)len << 16));
                    ticket[u] = atomicAdd(s_bucket + min(len, kLenBuckets) - 1, 1u);
                    mylen += (uint32_t)len;
                }
            }
        }
        if (__ballot(toolong) != 0ull && lane == 0) atomicOr(s_qn + 2, 1u);
        {  // exact number of samples this round can add, and the bound of its upstream gradients: LDS atomics by one lane
           // per wave (sample counts < 2^24: exact in float)
            // (the bound is the maximum over EVERY packet the round's records name, whether its rays turn out live or not, and
            //  whichever wave looks at it: single-round primitives walk their records in the order the forward appended them,
            //  which differs from run to run -- a bound that depended on which wave found live rays would make the scale, hence
            //  the bits of the slab gradient, depend on that order.  Found by test_full_batch_properties[C4] in round 6.)
            const float wl = wave_sum((float)mylen);
            if (lane == 0) {
                if (wl > 0.f) atomicAdd(s_qn + 1, (uint32_t)wl);
                if (gq_hi != 0u) atomicMax(s_gext, gq_hi);
            }
        }
        __syncthreads();
        // ---------------- this round's bound and scales (header) ----------------
        const uint32_t round_samples = (uint32_t)uni((int)s_qn[1]);  // exact; <= kQueueCap * 127 < 2^17
        const uint32_t gq_bits = (uint32_t)uni((int)s_gext[0]);
        const float Gq = __uint_as_float(gq_bits);
        bool bad_bound = false;
        float res_mul = 1.f;  // pass B: residuals (|r| <= 1/2) times this
        s_rgb = s_a = s_w = 1.f;
        if (round_samples > 0u && gq_bits != 0u) {  // (G_q = 0: every marched ray is skipped, scales are irrelevant)
            const float Brgb = wrgb * Gq, Ba = fa * Gq, Bw = WARP ? fw * Gq : 1.f;
            bad_bound = gq_bits >= 0x7f800000u || !(Brgb < 1.0e30f) || !(Ba < 1.0e30f) || !(Brgb > 1.0e-30f) ||
                        !(Ba > 1.0e-30f) || !(Bw < 1.0e30f);
            // (v_rcp_f32, 1 ulp: a scale need not be an exact quotient -- the sums are decoded with the reciprocal of the scale
            //  that was applied, and kFixRange keeps 0.1 % of headroom; five IEEE divisions per wave and round were 1.2 % of
            //  the kernel's instructions)
            res_mul = kFixRange * fast_rcp((float)round_samples);
            s_rgb = uni(res_mul * fast_rcp(Brgb)), s_a = uni(res_mul * fast_rcp(Ba));
            if constexpr (WARP) s_w = uni(res_mul * fast_rcp(fmaxf(Bw, 1.0e-30f)));
        }
        if constexpr (!RESID) {
            // (rare: header, ACCUMULATED ROUNDING) The noise budget kNoiseBudget * V bounds the sum over rounds of n_r^3; every
            // round may spend the share its entries have in the list.  A round over its share -- a box that is DEEP along its
            // rays: a dozen packets, thousands of samples each round -- is not marched: the rounds are cut shorter (n_r is
            // proportional to the entries taken, so e' = e * sqrt(share / n^3) entries fit) and this one starts again; the
            // entries are then walked in key order like those of any multi-round primitive, so every round, scale and flushed
            // float is still the same on every run.  (Round 6.  Before, such a primitive went to the two-pass instantiation:
            // 102 of 327 680 in the C2 training scene, marched twice by a small grid BEHIND this kernel -- 0.3-0.7 ms of serial
            // tail per backward)  Only a single entry over its share still does.
            const float n3c = (float)round_samples * (float)round_samples * (float)round_samples * (float)cnt;
            const float budget_e = kNoiseBudget * (float)V * (float)(eend - ebase);   // (n^3 > share  <=>  n^3 cnt > budget e)
            if (n3c > budget_e && s_qn[2] == 0u && !bad_bound) {
                if (eend - ebase > 1u) {  // (workgroup-uniform)
                    epr = max(1u, min(eend - ebase - 1u, (uint32_t)((float)(eend - ebase) * __builtin_sqrtf(budget_e / n3c))));
                    __syncthreads();  // (everybody has read this attempt's counters before the next one clears them)
                    if (!multi) {  // (then this is the first round, ebase = 0: nothing marched, nothing flushed)
                        multi = true;
                        // (the same ranks as rank_entries(), written as the plainest loop there is: this path is taken by a
                        //  handful of primitives per launch, and a second inlined copy of the blocked scalar-load form costs
                        //  every primitive 12 spilled SGPRs)
                        for (uint32_t e = tid; e < cnt; e += kPrimBlock) {
                            const uint32_t key = list[e].x;
                            uint32_t rank = 0u;
                            for (uint32_t j = 0; j < cnt; ++j) rank += (__builtin_nontemporal_load(&list[j].x) < key) ? 1u : 0u;
                            s_perm[rank] = (uint16_t)e;
                        }
                    }
                    continue;
                }
                if (tid == 0) {
                    atomicOr(p.pl_count + pk, kCountPrecise);
                    raise_flag(tail, kFlagBwdPrecise);
                }
                return;
            }
        }
        // a ray crosses this box over more than 127 steps, a sample weight left its bound in an earlier round (signed
        // opacity), or the upstream gradient / the slab is not finite: not this kernel's case
        if (s_qn[2] != 0u || bad_bound) {
            want_ray_centric();
            if (tid == 0) {
                atomicOr(p.pl_count + pk, kCountDead);
                raise_flag(tail, kFlagBwdHandoff);
            }
            // (addresses re-derived from a laundered pk: this exit sits inside the march loop and would otherwise keep
            //  the output pointers of the zero-fill live -- and spilled -- through the whole loop)
            size_t pkz = pk;
            int tz = tid;
            asm volatile("; zero-fill exit" : "+s"(pkz), "+v"(tz));
            float4 *gz = reinterpret_cast<float4 *>(p.grad_tplate) + pkz * (size_t)V;
            if constexpr (WARP) {
                float *gW = p.grad_warp + pkz * (size_t)VW * 3;
                for (int v = tz; v < VW * 3; v += kPrimBlock) gW[v] = 0.f;
            }
            for (int v = tz; v < V; v += kPrimBlock) gz[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tz < 9) p.grad_primrot[pkz * 9 + tz] = 0.f;
            if (tz < 3) p.grad_primscale[pkz * 3 + tz] = 0.f;
            if (tz < 3) p.grad_primpos[pkz * 3 + tz] = 0.f;
            return;
        }
        if (wave == 0) {  // exclusive prefix over the buckets, longest rays first (lane j <-> bucket kLenBuckets-1-j)
            const bool mine = lane < kLenBuckets;
            const uint32_t c = mine ? s_bucket[kLenBuckets - 1 - lane] : 0u;
            uint32_t incl = c;
#pragma unroll
            for (int d = 1; d < kLenBuckets; d <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, d);
                if (lane >= d) incl += up;
            }
            if (mine) s_bucket[kLenBuckets - 1 - lane] = incl - c;
            if (lane == kLenBuckets - 1) *s_qn = incl;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kEntriesPerWave; ++u)
            if (live2[u]) s_q[s_bucket[min((int)(item[u].y >> 16), kLenBuckets) - 1] + ticket[u]] = item[u];
        __syncthreads();
        // ---------------- phase 2: the queued rays in chunks of 64, dealt to the 4 waves ----------------
        // Full chunks, even when that leaves waves without work: the longest ray sets the round's critical path either
        // way, and 2 waves x 64 lanes issue half the instructions (VALU and LDS atomics) of 4 waves x 32 lanes.
        const int nq = (int)*s_qn;
        const int per = kWave;
        cur_mul = pass_b ? res_mul : 1.f;
        uint32_t gq_lo = 0x7fffffffu;  // bits of the smallest max |grad_rayrgba| over the rays this lane marches
        for (int qb = wave * per; qb < nq; qb += kPrimWaves * per) {
            // queue neighbours are usually neighbouring pixels, i.e. rays in the same slab cell: put them in DIFFERENT
            // 32-lane halves so that their LDS atomics to the same address do not meet in one pass
            const int ql = ((lane & 31) << 1) | (lane >> 5);
            const bool have = ql < per && qb + ql < nq;
            const uint2 it = have ? s_q[qb + ql] : make_uint2(0u, 0u);
            const uint32_t r = it.x & 0x7fffffu;  // index inside image n
            const int slo = (int)(it.y & 0xffffu);
            int len = have ? (int)(it.y >> 16) : 0;
            const uint32_t slot = it.x >> 23;
            f3 o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 1.f);
            float tmin = 0.f;
            f3 dL3 = mk3(0.f, 0.f, 0.f), rsat = mk3(-1.f, -1.f, -1.f);
            float dLw = 0.f, wbefore = 0.f, tend = -INFINITY;
            uint32_t satkey = 0u;
            if (have) {
                o = ld3(at_bytes<float>(raypos_n, r * 12u));
                d = ld3(at_bytes<float>(raydir_n, r * 12u));
                tmin = *at_bytes<float>(tminmax_n, r * 8u);
                const float4 g4 = *at_bytes<float4>(grad_n, r * 16u);
                dL3 = mk3(g4.x, g4.y, g4.z);
                dLw = g4.w;
                // bits of this ray's max |upstream gradient| (non-negative floats order like uints).  Zero: the ray adds
                // exact zeros to every gradient -- not marched
                const uint32_t gb = max(max(__float_as_uint(g4.x) & 0x7fffffffu, __float_as_uint(g4.y) & 0x7fffffffu),
                                        max(__float_as_uint(g4.z) & 0x7fffffffu, __float_as_uint(g4.w) & 0x7fffffffu));
                if (gb == 0u) len = 0; else gq_lo = min(gq_lo, gb);
                rsat = ld3(at_bytes<float>(raysat_n, r * 12u));
                const uint4 aux = *at_bytes<uint4>(aux_n, r * 16u);
                satkey = aux.x;
                wbefore = __uint_as_float(aux.y);
                tend = __uint_as_float(aux.w);
            }
            const bool has_sat = rsat.x > -1.f;  // primaccum.h:93
            const int nsteps = uni(wave_max(len));
            float ra0 = 0.f, ra1 = 0.f, ra2 = 0.f, rb0 = 0.f, rb1 = 0.f, rb2 = 0.f;
            // Queue neighbours are neighbouring pixels: at the same step index they sit in the same slab cell and their
            // 64 atomics hit the same addresses (serialised by the LDS).  The samples of a ray are independent here
            // (the forward recorded where the ray saturated), so each lane walks its steps from a different starting
            // offset, wrapping around: neighbours are then at different depths at any one time.
            int rot = have ? (int)(((uint32_t)ql * kRotMul) & kRotMask) : 0;
            while (rot >= len && len > 0) rot -= len;
            // The box coordinate is affine in the lattice step: y(s) = y(0) + s * dy, y(0) = box(o + d * tmin),
            // dy = (R^T d) * scale * dt -- two transforms per RAY instead of one per SAMPLE (3 fma instead of ~18 VALU).
            // (o and d themselves are needed again only after the walk, for the pose sums: re-read there, so that the
            //  walk carries 6 registers, not 12.)
            f3 ybase, dy;
            {
                const f3 x0 = mk3(fmaf(d.x, tmin, o.x), fmaf(d.y, tmin, o.y), fmaf(d.z, tmin, o.z));
                ybase = rot_rows(q, x0 - q.pos) * q.scale;
                dy = rot_rows(q, d) * q.scale * dt;
            }
            for (int st = 0; st < nsteps; ++st) {
                const int so = st + rot;
                const int s = slo + (so >= len ? so - len : so);
                const float sf = (float)s;
                const float t = fmaf(sf, dt, tmin);
                const f3 y = mk3(fmaf(sf, dy.x, ybase.x), fmaf(sf, dy.y, ybase.y), fmaf(sf, dy.z, ybase.z));
                const uint32_t key = ((uint32_t)s << 9) | slot;
                const bool inside = st < len && t < tend && key <= satkey && y.x > -1.f && y.x < 1.f && y.y > -1.f &&
                                    y.y < 1.f && y.z > -1.f && y.z < 1.f;
                if (__ballot(inside) == 0ull) continue;
                if constexpr (WARP) {
                    // ---- warp-field sampler (primsampler.h:68-91 with dowarp; utils.h:504-643 twice) ----
                    // Round 6: branch-free.  The first form guarded each of the 16 corners with its own bounds test (one exec
                    // region, one LDS read and one wait each) and scattered 32 + 24 one-word atomics; here every corner is read --
                    // a corner the zero padding leaves out has the weight 0 and reads a cell the sample reads anyway
                    // (march_common.h: tri_zero_pad) --, the arithmetic runs on the register pairs of the plain sampler and the
                    // sums are packed two per 64-bit atomic: (r | g), (b | a) per slab cell, (x | y) + z per warp node:
                    // 16 + 16 LDS atomics per sample.
                    if (inside) {
                        float fade;
                        f3 ypow;
                        if (FADE8) {
                            const f3 y2 = y * y, y4 = y2 * y2;
                            fade = fast_exp2(nfs_log2e * (y4.x * y4.x + y4.y * y4.y + y4.z * y4.z));
                            ypow = y4 * y2 * y;
                        } else {
                            const f3 ay = mk3(fabsf(y.x), fabsf(y.y), fabsf(y.z));
                            fade = fast_exp(-p.fadescale * (fast_pow(ay.x, p.fadeexp) + fast_pow(ay.y, p.fadeexp) +
                                                            fast_pow(ay.z, p.fadeexp)));
                            const float e1 = p.fadeexp - 1.f;
                            ypow = mk3(fast_pow(ay.x, e1) * (y.x > 0.f ? 1.f : -1.f),
                                       fast_pow(ay.y, e1) * (y.y > 0.f ? 1.f : -1.f),
                                       fast_pow(ay.z, e1) * (y.z > 0.f ? 1.f : -1.f));
                        }
                        // (1) y1 = warp(y).  y is strictly inside: every corner of the warp grid is in bounds once the base corner
                        //     is clamped (the plain sampler's form: one fma per index axis, float base corner, one conversion)
                        const float iwx = fmaf(y.x, mwx, mwx), iwy = fmaf(y.y, mwy, mwy), iwz = fmaf(y.z, mwz, mwz);
                        const float gx0 = fminf(floorf(iwx), (float)(WW - 2)), gy0 = fminf(floorf(iwy), (float)(WH - 2)),
                                    gz0 = fminf(floorf(iwz), (float)(WD - 2));
                        const float ux1 = iwx - gx0, uy1 = iwy - gy0, uz1 = iwz - gz0;
                        const v2f uxp = {1.f - ux1, ux1}, uyp = {1.f - uy1, uy1}, uzp = {1.f - uz1, uz1};
                        const int oH = WW, oD = WH * WW;
                        const int wvb = (int)fmaf(gz0, (float)oD, fmaf(gy0, (float)oH, gx0));
                        const v2f uyzA = pk_mul_lo(uyp, uzp), uyzB = pk_mul_hi(uyp, uzp);  // (uyz00, uyz10), (uyz01, uyz11)
                        const v2f U00 = pk_mul_lo(uxp, uyzA), U01 = pk_mul_hi(uxp, uyzA);
                        const v2f U10 = pk_mul_lo(uxp, uyzB), U11 = pk_mul_hi(uxp, uyzB);
                        const float4 *Wp = s_W + wvb;
#define MVP_LOADN(NAME_, IDX_)            \
    const float4 NAME_##q = Wp[IDX_];     \
    const v2f NAME_##l = {NAME_##q.x, NAME_##q.y};
                        MVP_LOADN(n000, 0)
                        MVP_LOADN(n001, 1)
                        MVP_LOADN(n010, oH)
                        MVP_LOADN(n011, oH + 1)
                        MVP_LOADN(n100, oD)
                        MVP_LOADN(n101, oD + 1)
                        MVP_LOADN(n110, oD + oH)
                        MVP_LOADN(n111, oD + oH + 1)
#undef MVP_LOADN
                        v2f y1l = pk_mul_lo(n000l, U00);
                        y1l = pk_fma_hi(n001l, U00, y1l), y1l = pk_fma_lo(n010l, U01, y1l), y1l = pk_fma_hi(n011l, U01, y1l);
                        y1l = pk_fma_lo(n100l, U10, y1l), y1l = pk_fma_hi(n101l, U10, y1l), y1l = pk_fma_lo(n110l, U11, y1l);
                        y1l = pk_fma_hi(n111l, U11, y1l);
                        float y1z = n000q.z * U00.x;
                        y1z = fmaf(n001q.z, U00.y, y1z), y1z = fmaf(n010q.z, U01.x, y1z), y1z = fmaf(n011q.z, U01.y, y1z);
                        y1z = fmaf(n100q.z, U10.x, y1z), y1z = fmaf(n101q.z, U10.y, y1z), y1z = fmaf(n110q.z, U11.x, y1z);
                        y1z = fmaf(n111q.z, U11.y, y1z);
                        // (2) the slab at y1 with the reference's zero padding (y1 may leave the slab)
                        const TriZ tt = tri_zero_pad(mk3(y1l.x, y1l.y, y1z), TD, TH, TW);
#define MVP_LOADC(NAME_, IDX_)              \
    const float4 NAME_##q = s_T[IDX_];      \
    const v2f NAME_##l = {NAME_##q.x, NAME_##q.y}, NAME_##h = {NAME_##q.z, NAME_##q.w};
                        MVP_LOADC(c000, tt.c[0])
                        MVP_LOADC(c001, tt.c[1])
                        MVP_LOADC(c010, tt.c[2])
                        MVP_LOADC(c011, tt.c[3])
                        MVP_LOADC(c100, tt.c[4])
                        MVP_LOADC(c101, tt.c[5])
                        MVP_LOADC(c110, tt.c[6])
                        MVP_LOADC(c111, tt.c[7])
#undef MVP_LOADC
                        v2f vl = pk_mul_lo(c000l, tt.W00), vh = pk_mul_lo(c000h, tt.W00);
                        vl = pk_fma_hi(c001l, tt.W00, vl), vh = pk_fma_hi(c001h, tt.W00, vh);
                        vl = pk_fma_lo(c010l, tt.W01, vl), vh = pk_fma_lo(c010h, tt.W01, vh);
                        vl = pk_fma_hi(c011l, tt.W01, vl), vh = pk_fma_hi(c011h, tt.W01, vh);
                        vl = pk_fma_lo(c100l, tt.W10, vl), vh = pk_fma_lo(c100h, tt.W10, vh);
                        vl = pk_fma_hi(c101l, tt.W10, vl), vh = pk_fma_hi(c101h, tt.W10, vh);
                        vl = pk_fma_lo(c110l, tt.W11, vl), vh = pk_fma_lo(c110h, tt.W11, vh);
                        vl = pk_fma_hi(c111l, tt.W11, vl), vh = pk_fma_hi(c111h, tt.W11, vh);
                        // (a sample whose y1 has an axis entirely outside the slab has eight zero weights and -- this kernel
                        //  marches finite slabs only -- the value 0, like the reference's empty sum)
                        float4 v;
                        v.x = vl.x, v.y = vl.y, v.z = vh.x, v.w = vh.y;
                        const float alpha = v.w * fade;
                        const bool issat = key == satkey;
                        const float weight = issat ? (1.f - wbefore) : alpha * dt;
                        wbad = wbad || !(fabsf(weight) <= wmax);
                        float4 dLs;
                        dLs.x = weight * dL3.x;
                        dLs.y = weight * dL3.y;
                        dLs.z = weight * dL3.z;
                        dLs.w = issat ? 0.f
                                      : dt * ((v.x - (has_sat ? rsat.x : 0.f)) * dL3.x +
                                              (v.y - (has_sat ? rsat.y : 0.f)) * dL3.y +
                                              (v.z - (has_sat ? rsat.z : 0.f)) * dL3.z + (has_sat ? 0.f : dLw));
                        const float gf = -(p.fadescale * p.fadeexp) * alpha * dLs.w;
                        f3 gy = ypow * gf;
                        dLs.w *= fade;
                        const v2f dl = {dLs.x, dLs.y}, dh = {dLs.z, dLs.w};
#define MVP_DOT4(NAME_, C_)                                 \
    float NAME_;                                            \
    {                                                       \
        const v2f p_ = C_##l * dl + C_##h * dh;             \
        NAME_ = p_.x + p_.y;                                \
    }
                        MVP_DOT4(d000, c000)
                        MVP_DOT4(d001, c001)
                        MVP_DOT4(d010, c010)
                        MVP_DOT4(d011, c011)
                        MVP_DOT4(d100, c100)
                        MVP_DOT4(d101, c101)
                        MVP_DOT4(d110, c110)
                        MVP_DOT4(d111, c111)
#undef MVP_DOT4
                        const f3 gi1 = posgrad_zero_pad(tt, d000, d001, d010, d011, d100, d101, d110, d111);
// two sums per 64-bit atomic: word = hi * 2^32 + lo (signed lo).  Pass B of a two-pass round adds what pass A rounded away.
#define MVP_PACK2W(LO_, HI_) ((unsigned long long)(uint32_t)(LO_) | ((unsigned long long)(uint32_t)((HI_) + ((LO_) >> 31)) << 32))
#define MVP_FIXP(PTR_, VLO_, VHI_)                                          \
    {                                                                       \
        const float xl_ = (VLO_), xh_ = (VHI_);                             \
        int lo_ = fix_rn(xl_), hi_ = fix_rn(xh_);                           \
        if (RESID && pass_b) {                                              \
            lo_ = fix_rn((xl_ - (float)lo_) * res_mul);                     \
            hi_ = fix_rn((xh_ - (float)hi_) * res_mul);                     \
        }                                                                   \
        atomicAdd((PTR_), MVP_PACK2W(lo_, hi_));                            \
    }
#define MVP_FIXS(PTR_, V_)                                                  \
    {                                                                       \
        const float x_ = (V_);                                              \
        int t_ = fix_rn(x_);                                                \
        if (RESID && pass_b) t_ = fix_rn((x_ - (float)t_) * res_mul);       \
        atomicAdd((PTR_), t_);                                              \
    }
                        {  // utils.h:582-589: the slab's share, in fixed point
                            const v2f qxy = {dLs.x * s_rgb, dLs.y * s_rgb}, qzw = {dLs.z * s_rgb, dLs.w * s_a};
                            unsigned long long *Ap = reinterpret_cast<unsigned long long *>(s_acc);
#define MVP_WSCATTER(CELL_, MUL_, WP_)                              \
    {                                                               \
        const v2f a_ = MUL_(qxy, WP_), b_ = MUL_(qzw, WP_);         \
        MVP_FIXP(Ap + (CELL_), a_.x, a_.y)                          \
        MVP_FIXP(Ap + Vp + (CELL_), b_.x, b_.y)                     \
    }
                            MVP_WSCATTER(tt.c[0], pk_mul_lo, tt.W00)
                            MVP_WSCATTER(tt.c[1], pk_mul_hi, tt.W00)
                            MVP_WSCATTER(tt.c[2], pk_mul_lo, tt.W01)
                            MVP_WSCATTER(tt.c[3], pk_mul_hi, tt.W01)
                            MVP_WSCATTER(tt.c[4], pk_mul_lo, tt.W10)
                            MVP_WSCATTER(tt.c[5], pk_mul_hi, tt.W10)
                            MVP_WSCATTER(tt.c[6], pk_mul_lo, tt.W11)
                            MVP_WSCATTER(tt.c[7], pk_mul_hi, tt.W11)
#undef MVP_WSCATTER
                        }
                        // (3) dL/dy1, the warp grid's share of it, and dL/dy through the warp lookup
                        const f3 g1 = mk3(mx * gi1.x, my * gi1.y, mz * gi1.z);
                        {
                            const v2f q1 = {g1.x * s_w, g1.y * s_w};
                            const float q1z = g1.z * s_w;
                            unsigned long long *Bp = reinterpret_cast<unsigned long long *>(s_wacc) + wvb;
                            int *Zp = s_wacc + 2 * VWp + wvb;
#define MVP_NSCATTER(OFF_, MUL_, WP_, WS_)                          \
    {                                                               \
        const v2f a_ = MUL_(q1, WP_);                               \
        MVP_FIXP(Bp + (OFF_), a_.x, a_.y)                           \
        MVP_FIXS(Zp + (OFF_), q1z * (WS_))                          \
    }
                            MVP_NSCATTER(0, pk_mul_lo, U00, U00.x)
                            MVP_NSCATTER(1, pk_mul_hi, U00, U00.y)
                            MVP_NSCATTER(oH, pk_mul_lo, U01, U01.x)
                            MVP_NSCATTER(oH + 1, pk_mul_hi, U01, U01.y)
                            MVP_NSCATTER(oD, pk_mul_lo, U10, U10.x)
                            MVP_NSCATTER(oD + 1, pk_mul_hi, U10, U10.y)
                            MVP_NSCATTER(oD + oH, pk_mul_lo, U11, U11.x)
                            MVP_NSCATTER(oD + oH + 1, pk_mul_hi, U11, U11.y)
#undef MVP_NSCATTER
                        }
#undef MVP_FIXS
#undef MVP_FIXP
#undef MVP_PACK2W
                        {  // d(warp lookup)/d(index) for the node values dotted with dL/dy1: the plain sampler's tree
#define MVP_DOT3(NAME_, N_) const float NAME_ = fmaf(N_##q.z, g1.z, fmaf(N_##q.y, g1.y, N_##q.x * g1.x));
                            MVP_DOT3(e000, n000)
                            MVP_DOT3(e001, n001)
                            MVP_DOT3(e010, n010)
                            MVP_DOT3(e011, n011)
                            MVP_DOT3(e100, n100)
                            MVP_DOT3(e101, n101)
                            MVP_DOT3(e110, n110)
                            MVP_DOT3(e111, n111)
#undef MVP_DOT3
                            const float dx00 = e001 - e000, dx10 = e011 - e010, dx01 = e101 - e100, dx11 = e111 - e110;
                            const float gix = fmaf(uyzB.y, dx11, fmaf(uyzB.x, dx01, fmaf(uyzA.y, dx10, uyzA.x * dx00)));
                            const float f00 = fmaf(ux1, dx00, e000), f10 = fmaf(ux1, dx10, e010);  // (y0,z0) (y1,z0)
                            const float f01 = fmaf(ux1, dx01, e100), f11 = fmaf(ux1, dx11, e110);  // (y0,z1) (y1,z1)
                            const float dy0 = f10 - f00, dy1 = f11 - f01;
                            const float giy = fmaf(uz1, dy1, uzp.x * dy0);
                            const float giz = fmaf(uy1, dy1, f01) - fmaf(uy1, dy0, f00);
                            gy.x = fmaf(mwx, gix, gy.x), gy.y = fmaf(mwy, giy, gy.y), gy.z = fmaf(mwz, giz, gy.z);
                        }
                        ra0 += gy.x, ra1 += gy.y, ra2 += gy.z;
                        rb0 = fmaf(t, gy.x, rb0), rb1 = fmaf(t, gy.y, rb1), rb2 = fmaf(t, gy.z, rb2);
                    }
                    continue;
                }
                if (inside) {
                    float fade;
                    f3 ypow;
                    if (FADE8) {
                        const f3 y2 = y * y, y4 = y2 * y2;
                        fade = fast_exp2(nfs_log2e * (y4.x * y4.x + y4.y * y4.y + y4.z * y4.z));  // exp(-fadescale * sum y^8)
                        ypow = y4 * y2 * y;
                    } else {
                        const f3 ay = mk3(fabsf(y.x), fabsf(y.y), fabsf(y.z));
                        fade = fast_exp(-p.fadescale * (fast_pow(ay.x, p.fadeexp) + fast_pow(ay.y, p.fadeexp) +
                                                        fast_pow(ay.z, p.fadeexp)));
                        const float e1 = p.fadeexp - 1.f;
                        ypow = mk3(fast_pow(ay.x, e1) * (y.x > 0.f ? 1.f : -1.f),
                                   fast_pow(ay.y, e1) * (y.y > 0.f ? 1.f : -1.f),
                                   fast_pow(ay.z, e1) * (y.z > 0.f ? 1.f : -1.f));
                    }
                    // (y + 1) / 2 * (T - 1) as ONE fma per axis (the forward's three roundings are not needed here: the
                    //  gradient is that of the same trilinear polynomial, evaluated at a point 1 ulp away at most)
                    const float ix = fmaf(y.x, mx, mx), iy = fmaf(y.y, my, my), iz = fmaf(y.z, mz, mz);
                    // base corner kept in float, weights exact, ONE conversion per offset (see tri_setup_f)
                    const float fx0 = fminf(floorf(ix), (float)(TW - 2)), fy0 = fminf(floorf(iy), (float)(TH - 2)),
                                fz0 = fminf(floorf(iz), (float)(TD - 2));
                    const float wx1 = ix - fx0, wy1 = iy - fy0, wz1 = iz - fz0;
                    const v2f wxp = {1.f - wx1, wx1}, wyp = {1.f - wy1, wy1}, wzp = {1.f - wz1, wz1};  // (w_0, w_1) per axis
                    const float wz0 = wzp.x;
                    const float vbf = fmaf(fz0, (float)sD, fmaf(fy0, (float)sH, fx0));  // (small integers: exact; sW = 1)
                    const int vb = (int)vbf;
                    // Corner values are kept as the (x,y) / (z,w) register pairs the 16-byte LDS reads deliver, so that
                    // interpolation and the 8 dot products below are packed-fp32 instructions on natural pairs.
#define MVP_LOADC(NAME_, IDX_)              \
    const float4 NAME_##q = s_T[IDX_];      \
    const v2f NAME_##l = {NAME_##q.x, NAME_##q.y}, NAME_##h = {NAME_##q.z, NAME_##q.w};
                    MVP_LOADC(c000, vb)
                    MVP_LOADC(c001, vb + sW)
                    MVP_LOADC(c010, vb + sH)
                    MVP_LOADC(c011, vb + sH + sW)
                    MVP_LOADC(c100, vb + sD)
                    MVP_LOADC(c101, vb + sD + sW)
                    MVP_LOADC(c110, vb + sD + sH)
                    MVP_LOADC(c111, vb + sD + sH + sW)
#undef MVP_LOADC
                    // the eight corner weights as four natural pairs W_zy = (w_zy0, w_zy1): six packed multiplies, no (w, w) pairs
                    const v2f wyzA = pk_mul_lo(wyp, wzp), wyzB = pk_mul_hi(wyp, wzp);  // (wyz00, wyz10), (wyz01, wyz11)
                    const float wyz00 = wyzA.x, wyz10 = wyzA.y, wyz01 = wyzB.x, wyz11 = wyzB.y;
                    const v2f W00 = pk_mul_lo(wxp, wyzA), W01 = pk_mul_hi(wxp, wyzA);  // (w000, w001), (w010, w011)
                    const v2f W10 = pk_mul_lo(wxp, wyzB), W11 = pk_mul_hi(wxp, wyzB);  // (w100, w101), (w110, w111)
                    v2f vl = pk_mul_lo(c000l, W00), vh = pk_mul_lo(c000h, W00);
                    vl = pk_fma_hi(c001l, W00, vl), vh = pk_fma_hi(c001h, W00, vh);
                    vl = pk_fma_lo(c010l, W01, vl), vh = pk_fma_lo(c010h, W01, vh);
                    vl = pk_fma_hi(c011l, W01, vl), vh = pk_fma_hi(c011h, W01, vh);
                    vl = pk_fma_lo(c100l, W10, vl), vh = pk_fma_lo(c100h, W10, vh);
                    vl = pk_fma_hi(c101l, W10, vl), vh = pk_fma_hi(c101h, W10, vh);
                    vl = pk_fma_lo(c110l, W11, vl), vh = pk_fma_lo(c110h, W11, vh);
                    vl = pk_fma_hi(c111l, W11, vl), vh = pk_fma_hi(c111h, W11, vh);
                    float4 v;
                    v.x = vl.x, v.y = vl.y, v.z = vh.x, v.w = vh.y;
                    const float alpha = v.w * fade;
                    const bool issat = key == satkey;
                    const float weight = issat ? (1.f - wbefore) : alpha * dt;
                    wbad = wbad || !(fabsf(weight) <= wmax);  // outside the fixed-point bound (signed opacity) or NaN
                    float4 dLs;
                    dLs.x = weight * dL3.x;
                    dLs.y = weight * dL3.y;
                    dLs.z = weight * dL3.z;
                    dLs.w = issat ? 0.f
                                  : dt * ((v.x - (has_sat ? rsat.x : 0.f)) * dL3.x +
                                          (v.y - (has_sat ? rsat.y : 0.f)) * dL3.y +
                                          (v.z - (has_sat ? rsat.z : 0.f)) * dL3.z + (has_sat ? 0.f : dLw));
                    const float gf = -(p.fadescale * p.fadeexp) * alpha * dLs.w;
                    f3 gy = ypow * gf;
                    dLs.w *= fade;
                    const v2f dl = {dLs.x, dLs.y}, dh = {dLs.z, dLs.w};
#define MVP_DOT4(NAME_, C_)                                 \
    float NAME_;                                            \
    {                                                       \
        const v2f p_ = C_##l * dl + C_##h * dh;             \
        NAME_ = p_.x + p_.y;                                \
    }
                    MVP_DOT4(d000, c000)
                    MVP_DOT4(d001, c001)
                    MVP_DOT4(d010, c010)
                    MVP_DOT4(d011, c011)
                    MVP_DOT4(d100, c100)
                    MVP_DOT4(d101, c101)
                    MVP_DOT4(d110, c110)
                    MVP_DOT4(d111, c111)
#undef MVP_DOT4
                    // utils.h:592-642, d/d(position) of the trilinear form, as a lerp tree over the eight dotted corners: the x
                    // differences of the four (y, z) edges give d/dx and the edge values, their y differences d/dy, the last
                    // difference d/dz -- 22 instructions instead of the 35 of the three separate weighted sums (the same
                    // polynomial; the kernel is bound by its VALU instruction count, profiles/r04_backward_experiments.json)
                    {
                        const float dx00 = d001 - d000, dx10 = d011 - d010, dx01 = d101 - d100, dx11 = d111 - d110;
                        const float gix = fmaf(wyz11, dx11, fmaf(wyz01, dx01, fmaf(wyz10, dx10, wyz00 * dx00)));
                        const float e00 = fmaf(wx1, dx00, d000), e10 = fmaf(wx1, dx10, d010);  // (y0,z0) (y1,z0)
                        const float e01 = fmaf(wx1, dx01, d100), e11 = fmaf(wx1, dx11, d110);  // (y0,z1) (y1,z1)
                        const float dy0 = e10 - e00, dy1 = e11 - e01;
                        const float giy = fmaf(wz1, dy1, wz0 * dy0);
                        const float giz = fmaf(wy1, dy1, e01) - fmaf(wy1, dy0, e00);
                        gy.x = fmaf(mx, gix, gy.x), gy.y = fmaf(my, giy, gy.y), gy.z = fmaf(mz, giz, gy.z);
                    }
                    // ---- utils.h:582-589 scatter, in fixed point (see the header of this kernel) ----
                    {
                        // scaled by powers of two (exact); pairs, so that weight x pair is one packed multiply
                        const v2f qxy = {dLs.x * s_rgb, dLs.y * s_rgb}, qzw = {dLs.z * s_rgb, dLs.w * s_a};
                        const int gb = (int)fmaf(fz0, (float)(gD - sD), vbf);  // z0 * gD + y0 * gH + x0 (gH = sH)
                        unsigned long long *Ap = reinterpret_cast<unsigned long long *>(s_acc) + gb;
// two channels per 64-bit atomic: word = hi * 2^32 + lo (signed lo): low dword lo, high dword hi + (lo >> 31)
#define MVP_PACK2(LO_, HI_) ((unsigned long long)(uint32_t)(LO_) | ((unsigned long long)(uint32_t)((HI_) + ((LO_) >> 31)) << 32))
#define MVP_FIX1(OFF_, VLO_, VHI_)                                                    \
    {                                                                                 \
        const int lo_ = fix_rn(VLO_), hi_ = fix_rn(VHI_);                             \
        atomicAdd(Ap + (OFF_), MVP_PACK2(lo_, hi_));                                  \
    }
// pass B of a two-pass round: what pass A rounded away, x - rn(x) (exact in fp32), at res_mul units per unit
#define MVP_FIX1B(OFF_, VLO_, VHI_)                                                   \
    {                                                                                 \
        const float xl_ = (VLO_), xh_ = (VHI_);                                       \
        const int lo_ = fix_rn((xl_ - (float)fix_rn(xl_)) * res_mul), hi_ = fix_rn((xh_ - (float)fix_rn(xh_)) * res_mul); \
        atomicAdd(Ap + (OFF_), MVP_PACK2(lo_, hi_));                                  \
    }
#define MVP_LSCATTER(FIX_, OFF_, MUL_, WP_)                         \
    {                                                               \
        const v2f a_ = MUL_(qxy, WP_), b_ = MUL_(qzw, WP_);         \
        FIX_((OFF_), a_.x, a_.y)                                    \
        FIX_((OFF_) + Vp, b_.x, b_.y)                               \
    }
#define MVP_LSCATTER8(FIX_)                                         \
    MVP_LSCATTER(FIX_, 0, pk_mul_lo, W00)                           \
    MVP_LSCATTER(FIX_, 1, pk_mul_hi, W00)                           \
    MVP_LSCATTER(FIX_, gH, pk_mul_lo, W01)                          \
    MVP_LSCATTER(FIX_, gH + 1, pk_mul_hi, W01)                      \
    MVP_LSCATTER(FIX_, gD, pk_mul_lo, W10)                          \
    MVP_LSCATTER(FIX_, gD + 1, pk_mul_hi, W10)                      \
    MVP_LSCATTER(FIX_, gD + gH, pk_mul_lo, W11)                     \
    MVP_LSCATTER(FIX_, gD + gH + 1, pk_mul_hi, W11)
                        if (!RESID || !pass_b) {  // (workgroup-uniform; the residual scatter exists in RESID only)
                            MVP_LSCATTER8(MVP_FIX1)
                        } else {
                            MVP_LSCATTER8(MVP_FIX1B)
                        }
#undef MVP_LSCATTER8
#undef MVP_LSCATTER
#undef MVP_FIX1B
#undef MVP_FIX1
#undef MVP_PACK2
                    }
                    // xmt = (o - pos) + d * t is affine in t along this ray: keep sum(gy) and sum(t * gy) only
                    ra0 += gy.x, ra1 += gy.y, ra2 += gy.z;
                    rb0 = fmaf(t, gy.x, rb0), rb1 = fmaf(t, gy.y, rb1), rb2 = fmaf(t, gy.z, rb2);
                }
            }
            if (!pass_b) {  // sum xmt_i * gy_j over this ray's samples = (o_i - pos_i) * sum(gy_j) + d_i * sum(t * gy_j)
                uint32_t r2 = r;
                asm volatile("; ray record re-read after the walk" : "+v"(r2));
                if (have) {
                    o = ld3(at_bytes<float>(raypos_n, r2 * 12u));
                    d = ld3(at_bytes<float>(raydir_n, r2 * 12u));
                }
                const f3 om = o - q.pos;
                a0 += ra0, a1 += ra1, a2 += ra2;
                c00 += om.x * ra0 + d.x * rb0, c01 += om.x * ra1 + d.x * rb1, c02 += om.x * ra2 + d.x * rb2;
                c10 += om.y * ra0 + d.y * rb0, c11 += om.y * ra1 + d.y * rb1, c12 += om.y * ra2 + d.y * rb2;
                c20 += om.z * ra0 + d.z * rb0, c21 += om.z * ra1 + d.z * rb1, c22 += om.z * ra2 + d.z * rb2;
            }
        }
        {
            const uint32_t wlo = (uint32_t)wave_min((int)gq_lo);  // (bit patterns < 2^31: signed order)
            if (lane == 0) atomicMin(s_gext + 1, wlo);
        }
        __syncthreads();  // the queue is rewritten by the next round
        if constexpr (RESID) {
            if (!pass_b && round_samples > 0u) {
                // pass A's sums leave, and the SAME round is marched again -- phase 1 included, it is deterministic --
                // accumulating the residuals pass A rounded away
                flush_sums(fast_rcp(s_rgb), fast_rcp(s_a), fast_rcp(s_w));
                pass_b = true;
                continue;
            }
        } else if (__uint_as_float((uint32_t)uni((int)s_gext[1])) * kTwoPassRatio < Gq) {
            // (rare: header, DYNAMIC RANGE) the rays marched here are more than 256x below the round's bound: the two-pass
            // instantiation, launched behind this kernel, owns the primitive and overwrites every output
            if (tid == 0) {
                atomicOr(p.pl_count + pk, kCountPrecise);
                raise_flag(p.pl_count + (size_t)p.N * K, kFlagBwdPrecise);
            }
            return;
        }
        if (ebase + epr < cnt && round_samples > 0u) {  // more rounds follow
            flush_sums(fast_rcp(s_rgb * cur_mul), fast_rcp(s_a * cur_mul), fast_rcp(s_w * cur_mul));
        }
        pass_b = false;
        ebase += epr;
    }
    // ---- pose gradients: 12 sums per lane -> wave -> workgroup (primtransf.h:155-179) ----
    {
        // (all 64 lanes are enabled here: workgroup-uniform control flow; lanes without rays hold zeros)
        float sums[12] = {a0, a1, a2, c00, c01, c02, c10, c11, c12, c20, c21, c22};
        wave_sum12_lane63(sums);
        if (lane == 63) {
#pragma unroll
            for (int j = 0; j < 12; ++j) s_red[wave * 12 + j] = sums[j];
        }
        if (__ballot(wbad) != 0ull && lane == 0) atomicOr(s_qn + 2, 2u);
    }
    __syncthreads();
    // The output addresses below depend only on (n, k, tid); left alone, the compiler computes them at kernel entry
    // (they also serve the early-exit zero-fill) and carries 9 pointer pairs through the whole march -- which is what
    // pushed this kernel over its 168-VGPR budget into scratch.  Re-deriving them from a laundered copy of pk keeps
    // them out of the hot loop's live set.
    size_t pkl = pk;
    int tl = tid;
    asm volatile("; late address base" : "+s"(pkl), "+v"(tl));
    float4 *gT4l = reinterpret_cast<float4 *>(p.grad_tplate) + pkl * (size_t)V;
    if (s_qn[2] != 0u) {  // a weight left its bound in the last round: the ray-centric kernel (fp32 atomics) owns it
        want_ray_centric();
        if (tl == 0) {
            atomicOr(p.pl_count + pkl, kCountDead);
            raise_flag(p.pl_count + (size_t)p.N * K, kFlagBwdHandoff);
        }
        if constexpr (WARP) {
            float *gW = p.grad_warp + pkl * (size_t)VW * 3;
            for (int v = tl; v < VW * 3; v += kPrimBlock) gW[v] = 0.f;
        }
        for (int v = tl; v < V; v += kPrimBlock) gT4l[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tl < 9) p.grad_primrot[pkl * 9 + tl] = 0.f;
        if (tl < 3) p.grad_primscale[pkl * 3 + tl] = 0.f;
        if (tl < 3) p.grad_primpos[pkl * 3 + tl] = 0.f;
        return;
    }
    {  // the slab gradient, written exactly once: sum / scale (of the last round's last pass)
        const float i_rgb = fast_rcp(s_rgb * cur_mul), i_a = fast_rcp(s_a * cur_mul);
        for (int v = tl; v < V; v += kPrimBlock) {
            const int z = v / sD, rem = v - z * sD;
            const int gv = z * gD + rem;  // (y * TW + x) is the same in both layouts
            float4 g;
            int ia_, ib_, ic_, id_;
            acc_read4<true>(s_acc, Vp, gv, ia_, ib_, ic_, id_);
            g.x = (float)ia_ * i_rgb, g.y = (float)ib_ * i_rgb, g.z = (float)ic_ * i_rgb, g.w = (float)id_ * i_a;
            if (drained) {  // (workgroup-uniform) earlier flushes sit in grad_template already; same owner thread
                const float4 o_ = gT4l[v];
                g.x = o_.x + g.x, g.y = o_.y + g.y, g.z = o_.z + g.z, g.w = o_.w + g.w;
            }
            MVP_STREAM_STORE(gT4l + v, g);
        }
    }
    if constexpr (WARP) {  // grad_warp, written exactly once
        const float i_w = fast_rcp(s_w * cur_mul);
        float *gW = p.grad_warp + pkl * (size_t)VW * 3;
        const int sDw = WH * WW;
        for (int v = tl; v < VW; v += kPrimBlock) {
            const int z = v / sDw, rem = v - z * sDw;
            const int gv = z * gDw + rem;
            const long long w0 = reinterpret_cast<const long long *>(s_wacc)[gv];  // (x | y), like acc_read4
            const int ix_ = (int)w0, iy_ = (int)((w0 - (long long)ix_) >> 32), iz_ = s_wacc[2 * VWp + gv];
            float gx_ = (float)ix_ * i_w, gy_ = (float)iy_ * i_w, gz_ = (float)iz_ * i_w;
            if (drained) gx_ += gW[v * 3], gy_ += gW[v * 3 + 1], gz_ += gW[v * 3 + 2];
            gW[v * 3] = gx_, gW[v * 3 + 1] = gy_, gW[v * 3 + 2] = gz_;
        }
    }
    if (tl < 12) {
        float t_ = s_red[tl];
#pragma unroll
        for (int w = 1; w < kPrimWaves; ++w) t_ += s_red[w * 12 + tl];
        s_red[48 + tl] = t_;
    }
    __syncthreads();
    if (tl < 15) {
        const float *Rg = p.primrot + pkl * 9, *sg = p.primscale + pkl * 3;
        const float *A = s_red + 48, *C = s_red + 51;  // A[j] = sum gy_j ; C[i*3+j] = sum xmt_i * gy_j
        // No sample evaluated (all twelve sums are exact zeros): the reference adds nothing (primtransf.h:155-179 runs only
        // when a lane evaluated, subset_kernel.h:203-205) -- in particular not 0 * NaN for a primitive whose transform is
        // not finite and which therefore can never be sampled.
        bool any = false;
#pragma unroll
        for (int j = 0; j < 12; ++j) any = any || (s_red[48 + j] != 0.f);
        if (!any) {
            if (tl < 9) p.grad_primrot[pkl * 9 + tl] = 0.f;
            else if (tl < 12) p.grad_primscale[pkl * 3 + tl - 9] = 0.f;
            else p.grad_primpos[pkl * 3 + tl - 12] = 0.f;
        } else if (tl < 9) {
            p.grad_primrot[pkl * 9 + tl] = sg[tl % 3] * C[tl];  // xmt_i * (gy_j * s_j)
        } else if (tl < 12) {
            const int j = tl - 9;  // sum_i R[i][j] * C[i][j] = sum rxmt_j * gy_j
            p.grad_primscale[pkl * 3 + j] = Rg[j] * C[j] + Rg[3 + j] * C[3 + j] + Rg[6 + j] * C[6 + j];
        } else {
            const int ii = tl - 12;
            p.grad_primpos[pkl * 3 + ii] =
                -(Rg[ii * 3 + 0] * sg[0] * A[0] + Rg[ii * 3 + 1] * sg[1] * A[1] + Rg[ii * 3 + 2] * sg[2] * A[2]);
        }
    }
}

// One workgroup per (image, primitive): the grid of prim_of_block.
template <bool FADE8, int TS, int PW, bool WARP = false>
__global__ __launch_bounds__(PW * 64, WARP ? 2 : kBwdOcc) void bwd_prim_kernel(const MarchParams p) {
    extern __shared__ __attribute__((aligned(16))) float4 smem4[];
    bwd_prim_body<FADE8, TS, PW, WARP, false>(p, (int)blockIdx.x, smem4);
}

// The two-pass instantiation (general slab strides): a small persistent grid that walks the same block -> primitive
// map and works only on the primitives the kernel above marked; returns at once when it marked none.  kPreciseWaves waves per
// workgroup: the marked primitives are few and LARGE (tens of thousands of samples each, marched twice), one per workgroup
// at a time, so the kernel lasts as long as its largest primitive -- more waves on it, not more workgroups, shorten that.
constexpr int kPreciseWaves = 4;
template <bool FADE8, bool WARP>
__global__ __launch_bounds__(kPreciseWaves * 64, WARP ? 1 : 2) void bwd_prim_precise_kernel(const MarchParams p, const int total_blocks) {
    extern __shared__ __attribute__((aligned(16))) float4 smem4[];
    if ((p.pl_count[(size_t)p.N * p.K] & kFlagBwdPrecise) == 0u) return;
    // Which of this workgroup's blocks are marked: all its counters are looked at in ONE parallel sweep (the body's own test is a
    // dependent global load + a barrier per block: ~160 blocks x ~2 us per workgroup at C2 when one or two of them have work)
    constexpr int kTodo = 256;
    __shared__ int s_todo[kTodo];
    __shared__ int s_ntodo;
    if (threadIdx.x == 0) s_ntodo = 0;
    __syncthreads();
    for (int b = (int)blockIdx.x + (int)threadIdx.x * (int)gridDim.x; b < total_blocks; b += (int)(blockDim.x * gridDim.x)) {
        int n, k;
        if (!prim_of_block(p, b, n, k)) continue;
        const uint32_t c = p.pl_count[(size_t)n * p.K + k];
        if ((c & (kCountPrecise | kCountDead)) != kCountPrecise) continue;
        const int slot = atomicAdd(&s_ntodo, 1);
        if (slot < kTodo) s_todo[slot] = b;
    }
    __syncthreads();
    const int ntodo = s_ntodo;
    if (ntodo > kTodo) {  // (more marked blocks than the table holds: the plain walk)
        for (int b = (int)blockIdx.x; b < total_blocks; b += (int)gridDim.x) {
            bwd_prim_body<FADE8, 0, kPreciseWaves, WARP, true>(p, b, smem4);
            __syncthreads();
        }
        return;
    }
    for (int i = 0; i < ntodo; ++i) {
        bwd_prim_body<FADE8, 0, kPreciseWaves, WARP, true>(p, s_todo[i], smem4);
        __syncthreads();  // (the next primitive restages the LDS this one's last readers may still be in)
    }
}

}  // namespace mvp

extern "C" int mvp_march_backward(int N, int H, int W, int K, const float *raypos, const float *raydir,
                                  float stepsize, const float *tminmax, const float *nodeaabb,
                                  const float *primpos, const float *primrot, const float *primscale, int TD,
                                  int TH, int TW, const float *tplate, int WD, int WH, int WW, const float *warp,
                                  const float *raysat, const uint32_t *rayaux, uint32_t *primlist_count,
                                  const uint32_t *primlist, int primlist_cap, const float *grad_rayrgba,
                                  float *grad_primpos, float *grad_primrot, float *grad_primscale,
                                  float *grad_tplate, float *grad_warp, float fadescale, float fadeexp,
                                  uint32_t *diag, void *stream) {
    using namespace mvp;
    MarchParams p = {};
    p.N = N, p.H = H, p.W = W, p.K = K, p.TD = TD, p.TH = TH, p.TW = TW;
    p.WD = WD, p.WH = WH, p.WW = WW, p.warp = warp, p.grad_warp = grad_warp;
    if (warp && (WD < 2 || WH < 2 || WW < 2)) return MVP_ERR_UNSUPPORTED;
    if (warp && !grad_warp) return MVP_ERR_BADARG;
    p.stepsize = stepsize, p.fadescale = fadescale, p.fadeexp = fadeexp;
    p.raypos = raypos, p.raydir = raydir, p.tminmax = tminmax, p.nodeaabb = nodeaabb;
    p.primpos = primpos, p.primrot = primrot, p.primscale = primscale, p.tplate = tplate;
    p.raysat_in = raysat, p.grad_rayrgba = grad_rayrgba;
    p.grad_primpos = grad_primpos, p.grad_primrot = grad_primrot, p.grad_primscale = grad_primscale;
    p.grad_tplate = grad_tplate, p.diag = diag;
    p.rayaux = const_cast<uint32_t *>(rayaux), p.pl_count = primlist_count;
    p.pl_list = reinterpret_cast<uint4 *>(const_cast<uint32_t *>(primlist)), p.pl_cap = primlist_cap;
    int rc = march_common_checks(true, p);
    if (rc == 1) rc = MVP_OK;  // no rays: the gradients are still defined (all zero) -> fall through to the fill
    if (rc != MVP_OK) return rc;
    if (K == 0 || N == 0) return MVP_OK;  // empty gradient tensors: nothing to write
    if (!grad_primpos || !grad_primrot || !grad_primscale || !grad_tplate) return MVP_ERR_BADARG;
    if (!aligned16(grad_tplate)) return MVP_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const size_t V = (size_t)TD * TH * TW;
    const bool norays = (long long)N * H * W == 0;
    if (!norays && (!raysat || !grad_rayrgba || !aligned16(grad_rayrgba))) return MVP_ERR_BADARG;
    const size_t padz = warp ? 0 : kGradPadZ;  // (bwd_prim_body: kPadZ)
    const size_t Vp = (size_t)TD * ((size_t)TH * TW + padz);
    // float4 slab + [4][Vp] int32 + ray queue + reduce area (+ queue tail)
    // 2 or 3 waves per workgroup by the work a primitive has: ray packets per primitive (see the note at kEntriesPerWave)
    const int pw = ((long long)p.tiles_x * p.tiles_y * 4 > 5ll * K) ? 3 : 2;
    // + list indices by rank (2 bytes per list slot; used by primitives whose list needs more than one round)
    size_t lds = V * 16 + Vp * 16 + (size_t)prim_queue_cap(pw) * 8 + 64 * sizeof(float) + 16 + kLenBuckets * 4 +
                 (((size_t)(primlist_cap > 0 ? primlist_cap : 0) * 2 + 15) & ~(size_t)15);
    if (warp) {  // + the warp grid (float4 per node) and its [3][VWp] accumulators
        lds = (lds + 15) & ~(size_t)15;
        p.prim_lds_base = (int)lds;
        const size_t VW = (size_t)WD * WH * WW, VWp = (size_t)WD * ((size_t)WH * WW + padz);
        lds += VW * 16 + VWp * 12;
    }
#ifdef MVP_DEBUG_HOOKS
    if (const char *e = getenv("MVP_DEBUG_LDS_PAD")) lds += (size_t)atoi(e);  // occupancy experiments: fewer workgroups per CU
#endif
    const bool have_lists = rayaux && primlist_count && primlist && primlist_cap > 0;
    // (queue items carry the ray index inside the image in 23 bits)
    const bool prim_path = !norays && have_lists && lds <= 64 * 1024 && (long long)H * W <= (1ll << 23);
    const bool fade8 = fadeexp == 8.0f;
    if (!prim_path) {  // ray-centric backward owns everything: it accumulates, so zero-fill first
        hipError_t e = hipMemsetAsync(grad_tplate, 0, sizeof(float) * 4 * V * (size_t)N * K, st);
        if (e == hipSuccess) e = hipMemsetAsync(grad_primpos, 0, sizeof(float) * 3 * (size_t)N * K, st);
        if (e == hipSuccess) e = hipMemsetAsync(grad_primrot, 0, sizeof(float) * 9 * (size_t)N * K, st);
        if (e == hipSuccess) e = hipMemsetAsync(grad_primscale, 0, sizeof(float) * 3 * (size_t)N * K, st);
        if (e == hipSuccess && warp)
            e = hipMemsetAsync(grad_warp, 0, sizeof(float) * 3 * (size_t)WD * WH * WW * (size_t)N * K, st);
        if (e != hipSuccess) return (int)e;
        if (norays) return MVP_OK;
        p.fallback_all = 1;
    } else {
        const long long pb = prim_grid_blocks(p);
        if (pb > 0x7fffffffll) return MVP_ERR_UNSUPPORTED;
        // bounds for the fixed-point scales: per-packet max |grad_rayrgba| behind the tail of primlist_count (max |raysat|
        // is in the tail already, written by the forward); also clears what an earlier backward left behind
        hipLaunchKernelGGL(packetmax_kernel, dim3(256 * 8), dim3(256), 0, st,
                           reinterpret_cast<const float4 *>(grad_rayrgba), N, H, W, p.tiles_x, p.tiles_y,
                           p.pl_count + (size_t)N * K + 3, p.pl_count, (size_t)N * K, p.pl_count + (size_t)N * K);
        rc = launch_status();
        if (rc != MVP_OK) return rc;
        const dim3 grid((unsigned)pb), block((unsigned)pw * 64);
        const bool cube8 = TD == 8 && TH == 8 && TW == 8;  // the reference's slab size (and BASELINE's)
#define MVP_LAUNCH_PRIM(F8_, TS_)                                                                  \
    {                                                                                              \
        if (pw == 3)                                                                               \
            hipLaunchKernelGGL((bwd_prim_kernel<F8_, TS_, 3>), grid, block, lds, st, p);           \
        else                                                                                       \
            hipLaunchKernelGGL((bwd_prim_kernel<F8_, TS_, 2>), grid, block, lds, st, p);           \
    }
#define MVP_LAUNCH_PRIMW(F8_)                                                                      \
    {                                                                                              \
        if (pw == 3)                                                                               \
            hipLaunchKernelGGL((bwd_prim_kernel<F8_, 0, 3, true>), grid, block, lds, st, p);       \
        else                                                                                       \
            hipLaunchKernelGGL((bwd_prim_kernel<F8_, 0, 2, true>), grid, block, lds, st, p);       \
    }
        if (warp && fade8)
            MVP_LAUNCH_PRIMW(true)
        else if (warp)
            MVP_LAUNCH_PRIMW(false)
        else if (fade8 && cube8)
            MVP_LAUNCH_PRIM(true, 8)
        else if (fade8)
            MVP_LAUNCH_PRIM(true, 0)
        else if (cube8)
            MVP_LAUNCH_PRIM(false, 8)
        else
            MVP_LAUNCH_PRIM(false, 0)
#undef MVP_LAUNCH_PRIM
#undef MVP_LAUNCH_PRIMW
        rc = launch_status();
        if (rc != MVP_OK) return rc;
        {  // the two-pass instantiation for the primitives that kernel marked (heavy-tailed upstream gradients); exits at
           // once otherwise.  Same LDS layout with kPreciseWaves waves per workgroup.
            size_t lds2 = V * 16 + Vp * 16 + (size_t)prim_queue_cap(kPreciseWaves) * 8 + 64 * sizeof(float) + 16 + kLenBuckets * 4 +
                          (((size_t)primlist_cap * 2 + 15) & ~(size_t)15);
            MarchParams p2 = p;
            if (warp) {
                lds2 = (lds2 + 15) & ~(size_t)15;
                p2.prim_lds_base = (int)lds2;
                const size_t VW = (size_t)WD * WH * WW, VWp = (size_t)WD * ((size_t)WH * WW + padz);
                lds2 += VW * 16 + VWp * 12;
            }
            const dim3 g2((unsigned)(pb < 2048 ? pb : 2048)), b2(kPreciseWaves * 64);
            if (warp && fade8)
                hipLaunchKernelGGL((bwd_prim_precise_kernel<true, true>), g2, b2, lds2, st, p2, (int)pb);
            else if (warp)
                hipLaunchKernelGGL((bwd_prim_precise_kernel<false, true>), g2, b2, lds2, st, p2, (int)pb);
            else if (fade8)
                hipLaunchKernelGGL((bwd_prim_precise_kernel<true, false>), g2, b2, lds2, st, p2, (int)pb);
            else
                hipLaunchKernelGGL((bwd_prim_precise_kernel<false, false>), g2, b2, lds2, st, p2, (int)pb);
            rc = launch_status();
            if (rc != MVP_OK) return rc;
        }
        p.fallback_all = 0;
    }
    // ray-centric kernel: everything (fallback_all) or only what the forward flagged; exits at once when no flag
    int fb = p.total_packets;
    if (!p.fallback_all && fb > 256 * 16) fb = 256 * 16;  // persistent-style grid for the rarely-taken path
    const dim3 grid((unsigned)fb), block(kWave);
    if (warp) {
        if (fade8)
            hipLaunchKernelGGL((march_kernel<true, true, true>), grid, block, 0, st, p);
        else
            hipLaunchKernelGGL((march_kernel<true, false, true>), grid, block, 0, st, p);
    } else {
        if (fade8)
            hipLaunchKernelGGL((march_kernel<true, true, false>), grid, block, 0, st, p);
        else
            hipLaunchKernelGGL((march_kernel<true, false, false>), grid, block, 0, st, p);
    }
    return launch_status();
}
