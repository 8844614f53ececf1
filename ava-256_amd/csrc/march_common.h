// march_common.h -- Mixture-of-Volumetric-Primitives raymarch, forward and backward, hand-written for gfx950 (CDNA4).
//
// WHAT it computes (bit-for-bit the same sample set and composition order as the reference):
//   /root/reference/extensions/mvpraymarch/mvpraymarch_subset_kernel.h:7-100   (forward)
//   /root/reference/extensions/mvpraymarch/mvpraymarch_subset_kernel.h:102-216 (backward, forwarddir=true)
//   utils.h:719-815 (fixed-order BVH traversal, leaf test), primtransf.h:105-179 (SRT), primsampler.h:44-91 +
//   utils.h:408-643 (fade + channels-last trilinear), primaccum.h:37-98 (additive accumulation, raysat rule).
//   A ray's result is  sum over lattice steps t_s = tmin + s*dt (s >= floor((rtmin-tmin)/dt), t_s < rtmax+1e-5),
//   over listed primitives in DFS-leaf order, of the samples whose box coordinate is strictly inside (-1,1)^3,
//   composited front to back until alpha saturates.
//
// HOW it is organised here is NOT the reference's schedule (a warp walks the tree node by node, then every
// step tests every listed primitive).  On CDNA4:
//   * one wave64 = one 8x8 pixel packet (one workgroup = one wave, private LDS, no cross-wave barriers);
//   * BVH traversal is breadth-first with LANES OVER NODES: each lane tests one frontier node's AABB against
//     the packet's interval bounds (origin box x 1/dir box, conservative), survivors are compacted in
//     left-to-right order with ballot + popcount prefix sums.  log2(K)-6 dependent memory round trips per
//     packet instead of one per visited node;
//   * candidates are then tested EXACTLY per ray (the reference's leaf test, utils.h:744-761) with LANES OVER
//     RAYS, reading the 15-float SRT records staged once into LDS; this yields the per-ray march interval
//     and, per listed primitive, a packet-level lattice-step range [lo,hi];
//   * the march sweeps lattice steps and visits only (step, primitive) pairs whose range contains the step:
//     a ballot over the ranges (lanes over list slots) gives the active-slot mask, empty stretches are
//     skipped with one wave-min.  Positions are evaluated directly, x_s = o + d*(tmin + s*dt), instead of by
//     ~150 accumulated fp32 adds (utils: subset_kernel.h:95-96), which is closer to the fp64 truth;
//   * packets that miss everything exit after the first frontier round (ray compaction by ballot).
//   * the inside-test guarantees all 8 trilinear corners are in bounds, so the sampler needs no bounds checks.
//   * backward is PRIMITIVE-centric and atomic-free in HBM (bwd_prim_kernel below): the gradient of a sample
//     does not depend on the running alpha once the forward has recorded, per ray, WHICH sample saturated it
//     and the alpha just before (rayaux), so samples can be regrouped by primitive.  The forward appends every
//     (packet, list slot, step range) to a per-primitive list; one workgroup per primitive then stages that
//     primitive's slab in LDS, re-evaluates its samples ray packet by ray packet, accumulates the slab gradient
//     in FIXED POINT with LDS integer atomics (ds_add_u32; ds_add_f32 retires ~3 cycles per active lane on gfx950,
//     see bwd_prim_kernel) and writes it back ONCE with coalesced 16-byte stores -- no global atomics, no zero-fill
//     pass.  The ray-centric backward with global_atomic_add_f32 (march_kernel<true,*>)
//     is kept as the always-correct fallback for primitives whose list overflowed (device-side flag).
//
// This header: parameters, constants and the device arithmetic BOTH march kernels use (box transform, fade, trilinear
// set-up and interpolation, lattice-step ranges, the block -> work mappings).  march_packet.h is the ray-packet march
// (forward kernel and the ray-centric backward), march_bwd.hip the primitive-centric backward, march_fwd.hip /
// march_bwd.hip hold the C-ABI entry points, march_host.hip the host logic they share.
#pragma once
#include <stdlib.h>

#include <type_traits>
#include "mvp_device.h"
#include "mvp_host.h"

namespace mvp {

constexpr int kTile = 8;          // 8x8 pixels per wave
constexpr int kMaxList = 512;     // reference hit-list cap (mvpraymarch_kernel.cu:101, utils.h:779)
constexpr int kRecSlots = 64;  // SRT records staged in LDS (first 64 candidates); beyond: scalar global loads
constexpr int kStartDepth = 10;   // the BFS tests every node of this depth first (implicit frontier, <= 1024 nodes)
constexpr int kNoSlot = 255;
// Lane-independent forward sweep (see march_packet): per-ray crossing table in LDS, kFastCross rows of 64 lanes.
// Row index 31 is the null link, so a ray can hold at most min(kFastCross, 31) crossings; packets beyond any of the
// limits below are marched by the slot-synchronous sweep instead (same results, slower).  kFastSlots records (64 B each)
// + kFastCross rows (256 B each) share the 8 KB the slot-synchronous layout needs: 40 + 22 keeps 5 waves per SIMD and
// measured best over C2/C3/C4 (64 + 24 at 10 KB, 48 + 20, 56 + 18 and 64 + 16 at 8 KB were within 3 %).
constexpr int kFastCross = 22;
constexpr int kFastMaxCross = kFastCross < 31 ? kFastCross : 31;
constexpr int kFastSlots = 40;  // list slots (6-bit field; every record of the fast path is LDS-resident) =
                                           // records staged in fast mode; the crossing table starts right behind them
constexpr int kFastCand = 128;    // BVH candidates (two registers per lane)
constexpr int kFastMaxLen = 64;   // lattice steps of one crossing (6-bit field)
constexpr int kFastMaxStep = 32767;  // largest lattice-step index (15-bit field)
constexpr uint32_t kNullLink = 31u;

struct MarchParams {
    int N, H, W, K;
    int TD, TH, TW;
    int tiles_x, tiles_y, chunk;  // 8x8 packets per image row / column; packets per (image, XCD) chunk
    float stepsize, fadescale, fadeexp;
    const float *raypos, *raydir, *tminmax, *nodeaabb, *primpos, *primrot, *primscale, *tplate;
    // forward only, instead of raypos/raydir/tminmax (all three null then): rays are made in the kernel from the cameras
    // with the arithmetic of raydirs_kernel (mvp_device.h: ray_from_camera) -- mvp_march_forward_cams
    const float *campos, *camrot, *focal, *princpt, *pixelcoords;
    float volradius;
    // ... and, in grad mode, written out for the backward (all three or none): what mvp_raydirs_forward would have written
    float *raypos_out, *raydir_out, *tminmax_out;
    int WD, WH, WW;                              // warp-field grid (algo 1), 0 when absent
    const float *warp;                           // [N,K,WD,WH,WW,3] or null
    float *grad_warp;                            // backward, algo 1
    float *rayrgba, *raysat;                     // forward outputs
    const float *raysat_in, *grad_rayrgba;       // backward inputs
    float *grad_primpos, *grad_primrot, *grad_primscale, *grad_tplate;
    uint32_t *diag;
    // forward -> backward hand-off (grad mode only; all may be null)
    uint32_t *rayaux;     // [N,H,W,4]: {satkey, bits(alpha before the saturating sample), first step, bits(tend)}
    uint32_t *pl_count;   // [N*K + 3 + N*tiles]: packets appended per primitive; flags (kFlag*), reserved, bits(Rmax);
                          // then per ray packet bits(max |grad_rayrgba|) of the current backward
    uint4 *pl_list;       // [N*K, pl_cap]: {(packet << 9) | list slot, lo | hi << 16, ray mask lo, ray mask hi}: the packet's
                          // step range in the primitive and which of its 64 rays have a lattice step there (round 6)
    int pl_cap;
    int fallback_all;     // backward: 1 = the ray-centric kernel handles every primitive
    int prim_lds_base;    // bwd_prim_kernel<.., WARP>: byte offset of the warp-field arrays in its dynamic LDS
    int total_packets;    // blocks of the march grid: images_whole * 8 * chunk + 8 * chunk * (N - images_whole)
    int images_whole;     // the first N - N % 8 images: XCD x owns images x, x + 8, ... whole
    int band_split;       // the other R = N % 8 images: F = band_split XCDs share each (2 for R = 4, 4 for R = 2, else 8),
    int band_chunk;       //   packet slots of one XCD's share; 8 / F images are in flight at a time
    // Only read by builds with -DMVP_DEBUG_HOOKS (tools/exp_variants.sh); the product library ignores the environment.
    int debug_force_dfs;  // MVP_DEBUG_FORCE_DFS=1 makes every packet take the exact DFS traversal
    int debug_slot_sweep; // MVP_DEBUG_SLOT_SWEEP=1 makes every packet take the slot-synchronous forward sweep
    int debug_stage;      // profiling (MVP_DEBUG_STAGE): 11 stop after the root test, 12 after the ancestor pre-cull,
                          // 13 after the implicit level, 1 after traversal, 2 after the exact pass, 3 no sampling
};
#ifdef MVP_DEBUG_HOOKS
#define MVP_DEBUG_STAGE(P_) ((P_).debug_stage)
#define MVP_DEBUG_FORCE_DFS(P_) ((P_).debug_force_dfs != 0)
#define MVP_DEBUG_SLOT_SWEEP(P_) ((P_).debug_slot_sweep != 0)
#else
#define MVP_DEBUG_STAGE(P_) 0
#define MVP_DEBUG_FORCE_DFS(P_) false
#define MVP_DEBUG_SLOT_SWEEP(P_) false
#endif

constexpr uint32_t kFlagListOverflow = 1u;  // some primitive received more than pl_cap packets
constexpr uint32_t kFlagGlobal = 2u;        // a packet produced step indices that do not fit the packed keys
constexpr uint32_t kFlagBwdHandoff = 4u;    // THIS backward handed a primitive to the ray-centric kernel (cleared per call)
constexpr uint32_t kFlagBwdPrecise = 8u;    // THIS backward left a primitive to the two-pass kernel (cleared per call)
constexpr uint32_t kCountDead = 0x80000000u;  // pl_count bit 31: "handed over by this backward" (cleared per call)
constexpr uint32_t kCountPrecise = 0x40000000u;  // bit 30: "owned by the two-pass (residual) kernel in this backward"
constexpr uint32_t kCountMask = 0x3fffffffu;     // the packets the forward counted
// Per-packet word behind the tail of pl_count: bit 31 = the FORWARD could not append this packet to some primitive's list
// (capacity), bit 30 = THIS backward wants the ray-centric kernel to march the packet (it is on the list of a primitive
// that kernel owns; cleared per call), bits 29..0 = bits(max |grad_rayrgba| of the packet) >> 2, rounded up.
constexpr uint32_t kPacketFwdOverflow = 0x80000000u, kPacketBwdWanted = 0x40000000u, kPacketMaxMask = 0x3fffffffu;
constexpr uint32_t kNoSat = 0xffffffffu;
constexpr int kStripRows = 3;  // packet rows per dispatch strip (packet_of_block: packet -> (image, tile))

// Streaming traffic is marked non-temporal so that it does not push re-used lines out of the L2: in the backward a
// primitive's slab is read once and its gradient written once per launch (5.4 GB at C2) while the ray records the same
// workgroups gather are re-read by the ~7 primitives a ray crosses; in the forward the rays are read once and the
// hand-off records (raysat, rayaux) are not read again before the backward, while slab lines are shared by neighbouring
// packets.
typedef __attribute__((ext_vector_type(4))) float nt_f4;  // (the builtins take native vectors, not HIP's float4 class)
__device__ __forceinline__ float4 stream_load(const float4 *p) {
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stream_store(float4 *p, float4 g) {
    const nt_f4 v = {g.x, g.y, g.z, g.w};
    __builtin_nontemporal_store(v, reinterpret_cast<nt_f4 *>(p));
}
#define MVP_STREAM_LOAD(P_) stream_load(P_)
#define MVP_STREAM_STORE(P_, V_) stream_store((P_), (V_))
#define MVP_STREAM_LOADF(P_) __builtin_nontemporal_load(P_)
#define MVP_STREAM_STOREF(P_, V_) __builtin_nontemporal_store((V_), (P_))

typedef float v2f __attribute__((ext_vector_type(2)));  // -> v_pk_mul_f32 / v_pk_fma_f32

// Raise flag bits in a shared word without queueing behind every other wave that raises the same bits (same-address
// atomics serialise in L2; a stale read only costs one redundant atomic)
__device__ __forceinline__ void raise_flag(uint32_t *word, uint32_t bits) {
    if ((__atomic_load_n(word, __ATOMIC_RELAXED) & bits) != bits) atomicOr(word, bits);
}

struct Rec {  // one primitive's transform, wave-uniform while it is being processed
    f3 pos, r0, r1, r2, scale;
};

// LDS image of a record, 4 x float4 per list slot, ordered so that the 16-byte reads deliver the register PAIRS the
// packed-fp32 box transform wants:  (r0.x r0.y r1.x r1.y) (r2.x r2.y pos.x pos.y) (r0.z r1.z r2.z pos.z) (s.x s.y s.z 0)
// (the spare word carries the primitive index k, so a lane that picks up a record needs no second lookup)
__device__ __forceinline__ void rec_to_lds(float4 *s_rec, int slot, const Rec &q, int k) {
    s_rec[slot * 4 + 0] = make_float4(q.r0.x, q.r0.y, q.r1.x, q.r1.y);
    s_rec[slot * 4 + 1] = make_float4(q.r2.x, q.r2.y, q.pos.x, q.pos.y);
    s_rec[slot * 4 + 2] = make_float4(q.r0.z, q.r1.z, q.r2.z, q.pos.z);
    s_rec[slot * 4 + 3] = make_float4(q.scale.x, q.scale.y, q.scale.z, __int_as_float(k));
}
__device__ __forceinline__ Rec rec_from_lds(const float4 *s_rec, int slot) {
    const float4 a = s_rec[slot * 4 + 0], b = s_rec[slot * 4 + 1], c = s_rec[slot * 4 + 2], d = s_rec[slot * 4 + 3];
    Rec r;
    r.pos = mk3(b.z, b.w, c.w);
    r.r0 = mk3(a.x, a.y, c.x);
    r.r1 = mk3(a.z, a.w, c.y);
    r.r2 = mk3(b.x, b.y, c.z);
    r.scale = mk3(d.x, d.y, d.z);
    return r;
}
// The same record as register pairs: y = (R^T (x - pos)) * s in 10 VALU instructions (pk_add, sub, pk_mul, 2 pk_fma,
// mul, 2 fma, pk_mul, mul) instead of 18 + the moves the compiler needs to build pairs out of f3 members.
struct RecP {
    v2f r0xy, r1xy, r2xy, pxy, sxy;
    float r0z, r1z, r2z, pz, sz;
};
struct Y3 {
    v2f xy;
    float z;
};
__device__ __forceinline__ RecP recp_from_lds(const float4 *s_rec, int slot) {
    const float4 a = s_rec[slot * 4 + 0], b = s_rec[slot * 4 + 1], c = s_rec[slot * 4 + 2], d = s_rec[slot * 4 + 3];
    RecP r;
    r.r0xy = v2f{a.x, a.y}, r.r1xy = v2f{a.z, a.w}, r.r2xy = v2f{b.x, b.y}, r.pxy = v2f{b.z, b.w};
    r.r0z = c.x, r.r1z = c.y, r.r2z = c.z, r.pz = c.w;
    r.sxy = v2f{d.x, d.y}, r.sz = d.z;
    return r;
}
__device__ __forceinline__ RecP recp_of(const Rec &q) {
    RecP r;
    r.r0xy = v2f{q.r0.x, q.r0.y}, r.r1xy = v2f{q.r1.x, q.r1.y}, r.r2xy = v2f{q.r2.x, q.r2.y};
    r.pxy = v2f{q.pos.x, q.pos.y}, r.sxy = v2f{q.scale.x, q.scale.y};
    r.r0z = q.r0.z, r.r1z = q.r1.z, r.r2z = q.r2.z, r.pz = q.pos.z, r.sz = q.scale.z;
    return r;
}
// ---- arithmetic shared by the two forward sweeps -------------------------------------------------------------------
// Both march schedules of the forward (lane-independent and slot-synchronous) must give every ray the SAME bits: a
// packet picks one or the other by its size, and a ray's value may not depend on the packet it sits in.  hipcc
// contracts a*b+c into an FMA per call site, so the same inline function can round differently in two places (it
// did: ~1 ulp on a third of the rays).  Everything both sweeps evaluate per sample is therefore written with EXPLICIT
// fused operations under `fp contract(off)`: box transform, ray position, fade, trilinear weights and interpolation,
// compositing.
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float a) { return v2f{a, a}; }
// Packed multiply / fma with ONE element of a register pair broadcast to both lanes through the instruction's op_sel bits
// (VOP3P: op_sel picks the source half of the low result lane, op_sel_hi that of the high one).  `a * splat(w)` written in C++
// makes the compiler build a (w, w) pair with a v_mov per weight -- eight per sample in the backward's walk; with the weights
// kept as the NATURAL pairs (w_x0, w_x1) * w_yz that four packed multiplies deliver, no pair has to be built at all.
__device__ __forceinline__ v2f pk_mul_lo(v2f a, v2f w) {  // a * w.x
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(w));
    return r;
}
__device__ __forceinline__ v2f pk_mul_hi(v2f a, v2f w) {  // a * w.y
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(w));
    return r;
}
__device__ __forceinline__ v2f pk_fma_lo(v2f a, v2f w, v2f c) {  // a * w.x + c
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
    return r;
}
__device__ __forceinline__ v2f pk_fma_hi(v2f a, v2f w, v2f c) {  // a * w.y + c
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
    return r;
}

// primtransf.h:119-132 for a direction (no translation) and for a point
__device__ __forceinline__ Y3 box_dir(const RecP &q, v2f vxy, float vz) {
#pragma clang fp contract(off)
    Y3 y;
    y.xy = pk_fma(q.r2xy, splat(vz), pk_fma(q.r1xy, splat(vxy.y), q.r0xy * splat(vxy.x))) * q.sxy;
    y.z = __builtin_fmaf(q.r2z, vz, __builtin_fmaf(q.r1z, vxy.y, q.r0z * vxy.x)) * q.sz;
    return y;
}
__device__ __forceinline__ Y3 box_point(const RecP &q, v2f xxy, float xz) {
#pragma clang fp contract(off)
    return box_dir(q, xxy - q.pxy, xz - q.pz);
}
// x = o + d * t at lattice step s, t = tmin + s * dt  (the reference accumulates, subset_kernel.h:95-96)
__device__ __forceinline__ float lattice_t(int s, float dt, float tmin) { return __builtin_fmaf((float)s, dt, tmin); }
__device__ __forceinline__ void ray_point(v2f oxy, float oz, v2f dxy, float dz, float t, v2f &xxy, float &xz) {
    xxy = pk_fma(dxy, splat(t), oxy);
    xz = __builtin_fmaf(dz, t, oz);
}
// primaccum.h:63-79: returns true when this sample saturates the ray (contrib is what was added to alpha)
__device__ __forceinline__ bool composite(float4 &rgba, const float4 &v, float dt, float &contrib) {
#pragma clang fp contract(off)
    const float newalpha = __builtin_fmaf(v.w, dt, rgba.w);
    contrib = fminf(newalpha, 1.f) - rgba.w;
    rgba.x = __builtin_fmaf(v.x, contrib, rgba.x);
    rgba.y = __builtin_fmaf(v.y, contrib, rgba.y);
    rgba.z = __builtin_fmaf(v.z, contrib, rgba.z);
    rgba.w = rgba.w + contrib;
    return newalpha >= 1.f;
}
__device__ __forceinline__ bool strictly_inside(const Y3 &y) {  // primtransf.h:112-117
    return fabsf(y.xy.x) < 1.f && fabsf(y.xy.y) < 1.f && fabsf(y.z) < 1.f;
}
__device__ __forceinline__ Rec rec_from_global(const float *pp, const float *pr, const float *ps, int k) {
    Rec r;
    r.pos = ld3(pp + (size_t)k * 3);
    r.r0 = ld3(pr + (size_t)k * 9);
    r.r1 = ld3(pr + (size_t)k * 9 + 3);
    r.r2 = ld3(pr + (size_t)k * 9 + 6);
    r.scale = ld3(ps + (size_t)k * 3);
    return r;
}

// primtransf.h:119-132: xmt = x - pos; rxmt = R0*xmt.x + R1*xmt.y + R2*xmt.z; y = rxmt * scale
__device__ __forceinline__ f3 rot_rows(const Rec &r, f3 v) {
    return mk3(r.r0.x * v.x + r.r1.x * v.y + r.r2.x * v.z, r.r0.y * v.x + r.r1.y * v.y + r.r2.y * v.z,
               r.r0.z * v.x + r.r1.z * v.y + r.r2.z * v.z);
}

struct AxisBounds {  // wave-uniform interval description of one axis of the 64 rays
    float olo, ohi;  // origin interval
    float ilo, ihi;  // 1/dir interval
    int sgn;         // +1: every active dir component > 0, -1: every one < 0, 0: mixed / zero
};
struct PacketBounds {  // conservative culling only
    AxisBounds ax, ay, az;
    float tlo, thi;  // [min tmin, max tmax + 1e-5]
};

__device__ __forceinline__ void axis_clip(const AxisBounds &a, float bmin, float bmax, float &tn, float &tf) {
    if (a.sgn > 0) {
        const float u = bmin - a.ohi;  // smallest (bmin - o)
        const float v = bmax - a.olo;  // largest (bmax - o)
        tn = fmaxf(tn, u * (u >= 0.f ? a.ilo : a.ihi));
        tf = fminf(tf, v * (v >= 0.f ? a.ihi : a.ilo));
    } else if (a.sgn < 0) {
        const float u = bmax - a.olo;  // largest (bmax - o); 1/dir < 0
        const float w = bmin - a.ohi;  // smallest (bmin - o)
        tn = fmaxf(tn, u * (u >= 0.f ? a.ilo : a.ihi));
        tf = fminf(tf, w * (w <= 0.f ? a.ilo : a.ihi));
    }
}

// Conservative: returns true whenever ANY ray of the packet passes the reference's slab test
// (utils.h:679-685) within the packet's t range; extra candidates are harmless (exact test follows).
__device__ __forceinline__ bool packet_hits_box(const PacketBounds &pb, float x0, float y0, float z0, float x1,
                                                float y1, float z1) {
    float tn = pb.tlo, tf = pb.thi;
    axis_clip(pb.ax, x0, x1, tn, tf);
    axis_clip(pb.ay, y0, y1, tn, tf);
    axis_clip(pb.az, z0, z1, tn, tf);
    // A box whose six corner coordinates are all NaN (primscale = 0 under an axis-aligned rotation: inf * 0 in
    // primtransf.h:12-63) is never entered by the reference: max_component / min_component of three NaN axes are NaN and the
    // comparison fails (utils.h:659-665,679-685).  A NaN axis next to a valid one is ignored there, as the clips above do.
    const bool all_nan = (x0 != x0) && (x1 != x1) && (y0 != y0) && (y1 != y1) && (z0 != z0) && (z1 != z1);
    return tn <= tf + 1e-4f + 1e-5f * fabsf(tf) && !all_nan;
}

// same_o: every active ray of the packet starts at the same point (wave-uniform, decided once per packet): a pinhole
// camera's rays do (utils_kernel.cu:30-32: raypos = campos / volradius), so the origin interval is that point and six of the
// packet's fourteen wave reductions are not needed
__device__ __forceinline__ AxisBounds axis_bounds(bool active, float o, float d, bool same_o, float o_first) {
    AxisBounds a;
    if (same_o) {
        a.olo = a.ohi = o_first;
    } else {
        a.olo = uni(wave_min(active ? o : INFINITY));
        a.ohi = uni(wave_max(active ? o : -INFINITY));
    }
    const float ird = fast_rcp(d);  // (1 ulp; the bounds only cull, with the slack of packet_hits_box, and the exact test follows)
    a.ilo = uni(wave_min(active ? ird : INFINITY));
    a.ihi = uni(wave_max(active ? ird : -INFINITY));
    const bool allpos = __ballot(active && !(d > 0.f)) == 0ull;
    const bool allneg = __ballot(active && !(d < 0.f)) == 0ull;
    a.sgn = allpos ? 1 : (allneg ? -1 : 0);
    return a;
}

// One ray packet (8x8 pixels, one wave).  s_a: frontier ping, later packed step ranges (lo | hi << 16);
// s_b: frontier pong / candidate list / final list (k | slot << 24); s_rec: SRT records of the first 64 candidates.
// BWD instantiation = ray-centric fallback backward; emit_all: it owns every primitive (else only overflowed ones).
// Forward sample of one slab at box coordinate y (strictly inside (-1,1)^3): fade (primsampler.h:48-51) times the
// channels-last trilinear lookup (utils.h:414-468; base corner clamped so that all 8 corners are in bounds, which
// gives the same value as the reference's zero-padded form).  Returns (r, g, b, alpha * fade).
// fade = exp(-fadescale * sum |y_i|^fadeexp)  (primsampler.h:48-51)
template <bool FADE8>
__device__ __forceinline__ float fade_pinned(f3 y, float fadescale, float fadeexp) {
#pragma clang fp contract(off)
    float e;
    if (FADE8) {
        const f3 y2 = y * y, y4 = y2 * y2;
        e = __builtin_fmaf(y4.z, y4.z, __builtin_fmaf(y4.y, y4.y, y4.x * y4.x));
    } else {
        e = (fast_pow(fabsf(y.x), fadeexp) + fast_pow(fabsf(y.y), fadeexp)) + fast_pow(fabsf(y.z), fadeexp);
    }
    return fast_exp2((-1.44269504088896341f * fadescale) * e);  // the scale product is loop-invariant: one multiply per sample
}
struct Tri {  // base corner (clamped so that all 8 corners are in bounds) and the 8 corner weights
    int x0, y0, z0;
    float w000, w001, w010, w011, w100, w101, w110, w111;
};
__device__ __forceinline__ Tri tri_setup(f3 y, float mx, float my, float mz, int TW, int TH, int TD) {
#pragma clang fp contract(off)
    const float ix = ((y.x + 1.f) * 0.5f) * mx, iy = ((y.y + 1.f) * 0.5f) * my, iz = ((y.z + 1.f) * 0.5f) * mz;
    Tri t;
    t.x0 = min((int)floorf(ix), TW - 2), t.y0 = min((int)floorf(iy), TH - 2), t.z0 = min((int)floorf(iz), TD - 2);
    const float wx1 = ix - (float)t.x0, wx0 = (float)(t.x0 + 1) - ix;
    const float wy1 = iy - (float)t.y0, wy0 = (float)(t.y0 + 1) - iy;
    const float wz1 = iz - (float)t.z0, wz0 = (float)(t.z0 + 1) - iz;
    const float wyz00 = wy0 * wz0, wyz10 = wy1 * wz0, wyz01 = wy0 * wz1, wyz11 = wy1 * wz1;
    t.w000 = wx0 * wyz00, t.w001 = wx1 * wyz00, t.w010 = wx0 * wyz10, t.w011 = wx1 * wyz10;
    t.w100 = wx0 * wyz01, t.w101 = wx1 * wyz01, t.w110 = wx0 * wyz11, t.w111 = wx1 * wyz11;
    return t;
}
// The same weights with the base corner kept in FLOAT (compile-time slab size): float -> int conversions run at a quarter of
// the plain VALU rate on gfx950 (tools/ubench/valu_rate.hip: v_cvt_* ~5 cycles per wave instruction against ~3), and the
// integer form above needs nine of them per sample (three floor -> int, six int -> float for the weights).  Here the clamp
// is a float min, the weights are ix - fx0 and 1 - (ix - fx0) -- both EXACT, so the same bits as (x0 + 1) - ix: ix and
// fx0 <= ix are multiples of ulp(ix) -- and the cell's byte offset is formed in float (small exact integers) and converted
// ONCE.  Identical results, ~12 fewer instructions per sample, eight of them conversions.
struct TriF {
    uint32_t off;  // byte offset of the base corner inside a TS^3 float4 slab
    v2f W00, W01, W10, W11;  // the natural pairs W_zy = (w_zy0, w_zy1)
};
// Round 4: the index is ONE fma per axis (y * m/2 + m/2; the three-operation form (y + 1) * 0.5 * m it replaces differs by
// <= 1 ulp of the index, and both forward sweeps call this one function, so they still agree bit for bit), and the eight
// weights come out of six packed multiplies as the natural pairs (w_x0, w_x1) * w_yz -- the same products in the same order
// as the scalar form, so the weights themselves are unchanged.
template <int TS>
__device__ __forceinline__ TriF tri_setup_f(f3 y) {
#pragma clang fp contract(off)
    constexpr float h = 0.5f * (float)(TS - 1), top = (float)(TS - 2);
    const float ix = __builtin_fmaf(y.x, h, h), iy = __builtin_fmaf(y.y, h, h), iz = __builtin_fmaf(y.z, h, h);
    const float fx0 = fminf(floorf(ix), top), fy0 = fminf(floorf(iy), top), fz0 = fminf(floorf(iz), top);
    const float wx1 = ix - fx0, wy1 = iy - fy0, wz1 = iz - fz0;
    const v2f wxp{1.f - wx1, wx1}, wyp{1.f - wy1, wy1}, wzp{1.f - wz1, wz1};
    TriF t;
    t.off = (uint32_t)__builtin_fmaf(fz0, (float)(TS * TS * 16), __builtin_fmaf(fy0, (float)(TS * 16), fx0 * 16.f));
    const v2f wyzA = pk_mul_lo(wyp, wzp), wyzB = pk_mul_hi(wyp, wzp);  // (wyz00, wyz10), (wyz01, wyz11)
    t.W00 = pk_mul_lo(wxp, wyzA), t.W01 = pk_mul_hi(wxp, wyzA);        // (w000, w001), (w010, w011)
    t.W10 = pk_mul_lo(wxp, wyzB), t.W11 = pk_mul_hi(wxp, wyzB);        // (w100, w101), (w110, w111)
    return t;
}
// sum_c w_c * corner_c on the (x,y)/(z,w) register pairs the 16-byte loads deliver, in corner order 000,001,..,111
template <class TRI>
__device__ __forceinline__ float4 tri_interp(const TRI &t, const float4 &c000, const float4 &c001, const float4 &c010,
                                             const float4 &c011, const float4 &c100, const float4 &c101,
                                             const float4 &c110, const float4 &c111) {
#pragma clang fp contract(off)
#define MVP_L(C_) v2f{(C_).x, (C_).y}
#define MVP_H(C_) v2f{(C_).z, (C_).w}
    v2f vl = MVP_L(c000) * splat(t.w000), vh = MVP_H(c000) * splat(t.w000);
    vl = pk_fma(MVP_L(c001), splat(t.w001), vl), vh = pk_fma(MVP_H(c001), splat(t.w001), vh);
    vl = pk_fma(MVP_L(c010), splat(t.w010), vl), vh = pk_fma(MVP_H(c010), splat(t.w010), vh);
    vl = pk_fma(MVP_L(c011), splat(t.w011), vl), vh = pk_fma(MVP_H(c011), splat(t.w011), vh);
    vl = pk_fma(MVP_L(c100), splat(t.w100), vl), vh = pk_fma(MVP_H(c100), splat(t.w100), vh);
    vl = pk_fma(MVP_L(c101), splat(t.w101), vl), vh = pk_fma(MVP_H(c101), splat(t.w101), vh);
    vl = pk_fma(MVP_L(c110), splat(t.w110), vl), vh = pk_fma(MVP_H(c110), splat(t.w110), vh);
    vl = pk_fma(MVP_L(c111), splat(t.w111), vl), vh = pk_fma(MVP_H(c111), splat(t.w111), vh);
#undef MVP_L
#undef MVP_H
    return make_float4(vl.x, vl.y, vh.x, vh.y);
}

// The same sum on the natural weight pairs: the op_sel broadcast forms above, no (w, w) pair is ever built
__device__ __forceinline__ float4 tri_interp(const TriF &t, const float4 &c000, const float4 &c001, const float4 &c010,
                                             const float4 &c011, const float4 &c100, const float4 &c101,
                                             const float4 &c110, const float4 &c111) {
#define MVP_L(C_) v2f{(C_).x, (C_).y}
#define MVP_H(C_) v2f{(C_).z, (C_).w}
    v2f vl = pk_mul_lo(MVP_L(c000), t.W00), vh = pk_mul_lo(MVP_H(c000), t.W00);
    vl = pk_fma_hi(MVP_L(c001), t.W00, vl), vh = pk_fma_hi(MVP_H(c001), t.W00, vh);
    vl = pk_fma_lo(MVP_L(c010), t.W01, vl), vh = pk_fma_lo(MVP_H(c010), t.W01, vh);
    vl = pk_fma_hi(MVP_L(c011), t.W01, vl), vh = pk_fma_hi(MVP_H(c011), t.W01, vh);
    vl = pk_fma_lo(MVP_L(c100), t.W10, vl), vh = pk_fma_lo(MVP_H(c100), t.W10, vh);
    vl = pk_fma_hi(MVP_L(c101), t.W10, vl), vh = pk_fma_hi(MVP_H(c101), t.W10, vh);
    vl = pk_fma_lo(MVP_L(c110), t.W11, vl), vh = pk_fma_lo(MVP_H(c110), t.W11, vh);
    vl = pk_fma_hi(MVP_L(c111), t.W11, vl), vh = pk_fma_hi(MVP_H(c111), t.W11, vh);
#undef MVP_L
#undef MVP_H
    return make_float4(vl.x, vl.y, vh.x, vh.y);
}

template <bool FADE8>
__device__ __forceinline__ float4 sample_slab(const float *__restrict__ Tk, f3 y, int TD, int TH, int TW,
                                              float fadescale, float fadeexp) {
#pragma clang fp contract(off)
    const float fade = fade_pinned<FADE8>(y, fadescale, fadeexp);
    const Tri t = tri_setup(y, (float)(TW - 1), (float)(TH - 1), (float)(TD - 1), TW, TH, TD);
    const int sW = 4, sH = TW * 4, sD = TH * TW * 4;
    const float *Tp = Tk + (size_t)t.z0 * sD + (size_t)t.y0 * sH + (size_t)t.x0 * sW;
    const float4 c000 = *reinterpret_cast<const float4 *>(Tp);
    const float4 c001 = *reinterpret_cast<const float4 *>(Tp + sW);
    const float4 c010 = *reinterpret_cast<const float4 *>(Tp + sH);
    const float4 c011 = *reinterpret_cast<const float4 *>(Tp + sH + sW);
    const float4 c100 = *reinterpret_cast<const float4 *>(Tp + sD);
    const float4 c101 = *reinterpret_cast<const float4 *>(Tp + sD + sW);
    const float4 c110 = *reinterpret_cast<const float4 *>(Tp + sD + sH);
    const float4 c111 = *reinterpret_cast<const float4 *>(Tp + sD + sH + sW);
    float4 v = tri_interp(t, c000, c001, c010, c011, c100, c101, c110, c111);
    v.w = v.w * fade;
    return v;
}

// The same for a TS^3 slab with compile-time strides.  Timg = the image's template block (wave-uniform, SGPR pair),
// kbyte = byte offset of this lane's slab inside it (< 2^32, checked by the host): every gather is
// "scalar base + 32-bit lane offset + immediate", no 64-bit address arithmetic in the sweep.
template <bool FADE8, int TS>
__device__ __forceinline__ float4 sample_slab_c(const float *__restrict__ Timg, uint32_t kbyte, f3 y, float fadescale,
                                                float fadeexp) {
#pragma clang fp contract(off)
    const float fade = fade_pinned<FADE8>(y, fadescale, fadeexp);
    const TriF t = tri_setup_f<TS>(y);
    constexpr int bW = 16, bH = TS * 16, bD = TS * TS * 16;  // byte strides
    const uint32_t off = kbyte + t.off;
    const char *pc = reinterpret_cast<const char *>(Timg) + (size_t)off;
#define MVP_C(O_) (*reinterpret_cast<const float4 *>(pc + (O_)))
    const float4 c000 = MVP_C(0), c001 = MVP_C(bW), c010 = MVP_C(bH), c011 = MVP_C(bH + bW);
    const float4 c100 = MVP_C(bD), c101 = MVP_C(bD + bW), c110 = MVP_C(bD + bH), c111 = MVP_C(bD + bH + bW);
#undef MVP_C
    float4 v = tri_interp(t, c000, c001, c010, c011, c100, c101, c110, c111);
    v.w = v.w * fade;
    return v;
}

// ---- half-precision slabs (opt-in render path, round 5) -----------------------------------------------------------
// The forward's sweep is bound by the texture addresser, which charges per lane-gather: eight 16-byte gathers
// per sample.  With the slab stored as fp16 RGBA (8 bytes per voxel, 4 KB per 8^3 slab: mvp_template_to_half,
// mvp_template_assemble_forward_half) the two x-neighbours of a corner pair are 16 CONTIGUOUS bytes (8-byte aligned, never
// across a 64-byte slab row), so a sample is FOUR 16-byte gathers over half the bytes.  Weights, interpolation and
// compositing stay fp32: every product is v_fma_mix_f32 (fp16 operand widened inside the instruction, fp32 accumulate), in
// the corner order of the fp32 sampler.  The only difference from the fp32 path is the storage rounding of the slab itself
// (2^-11 relative per voxel value): against the oracle run ON THE ROUNDED SLABS the kernel meets the standing forward
// tolerance (tests/test_gpu_half.py).
typedef _Float16 h8 __attribute__((ext_vector_type(8), aligned(8)));
template <bool FADE8, int TS>
__device__ __forceinline__ float4 sample_slab_h(const void *__restrict__ Timg, uint32_t kbyte, f3 y, float fadescale,
                                                float fadeexp) {
#pragma clang fp contract(off)
    const float fade = fade_pinned<FADE8>(y, fadescale, fadeexp);
    const TriF t = tri_setup_f<TS>(y);                       // (t.off counts 16-byte voxels: half of it here)
    constexpr int bH = TS * 8, bD = TS * TS * 8;             // byte strides of the fp16 slab
    const char *pc = reinterpret_cast<const char *>(Timg) + (size_t)(kbyte + (t.off >> 1));
    // one 16-byte gather per (y, z) corner pair: [0..3] = rgba at x0, [4..7] = rgba at x0 + 1
    const h8 p00 = *reinterpret_cast<const h8 *>(pc), p01 = *reinterpret_cast<const h8 *>(pc + bH);
    const h8 p10 = *reinterpret_cast<const h8 *>(pc + bD), p11 = *reinterpret_cast<const h8 *>(pc + bD + bH);
    float4 v;
#define MVP_HSUM(C_)                                                                                          \
    __builtin_fmaf((float)p11[4 + C_], t.W11.y,                                                               \
    __builtin_fmaf((float)p11[C_], t.W11.x,                                                                   \
    __builtin_fmaf((float)p10[4 + C_], t.W10.y,                                                               \
    __builtin_fmaf((float)p10[C_], t.W10.x,                                                                   \
    __builtin_fmaf((float)p01[4 + C_], t.W01.y,                                                               \
    __builtin_fmaf((float)p01[C_], t.W01.x,                                                                   \
    __builtin_fmaf((float)p00[4 + C_], t.W00.y, __builtin_fmaf((float)p00[C_], t.W00.x, 0.f))))))))
    v.x = MVP_HSUM(0), v.y = MVP_HSUM(1), v.z = MVP_HSUM(2), v.w = MVP_HSUM(3);
#undef MVP_HSUM
    v.w = v.w * fade;
    return v;
}

// ---- warp-field path (algo 1: PrimSamplerTW<true>, primsampler.h:53-58,82-88) -------------------------------------
// The warped coordinate y1 may leave (-1,1)^3, so the template lookup needs the reference's general form: normalised
// coordinate clamped to +-100, floor, zero padding through per-corner bounds tests (utils.h:414-498).  This guarded form is
// what the ray-centric backward (march_packet.h) runs -- the always-correct owner, non-finite slabs included; the forward and
// the primitive-centric backward use the branch-free form below.
struct TriG {
    int x0, y0, z0;
    float wx0, wx1, wy0, wy1, wz0, wz1;
};
__device__ __forceinline__ TriG tri_general(f3 y, int D, int H, int W) {
    const float ix = fmaxf(-100.f, fminf(100.f, (y.x + 1.f) * 0.5f)) * (float)(W - 1);
    const float iy = fmaxf(-100.f, fminf(100.f, (y.y + 1.f) * 0.5f)) * (float)(H - 1);
    const float iz = fmaxf(-100.f, fminf(100.f, (y.z + 1.f) * 0.5f)) * (float)(D - 1);
    TriG t;
    t.x0 = (int)floorf(ix), t.y0 = (int)floorf(iy), t.z0 = (int)floorf(iz);
    t.wx1 = ix - (float)t.x0, t.wx0 = (float)(t.x0 + 1) - ix;
    t.wy1 = iy - (float)t.y0, t.wy0 = (float)(t.y0 + 1) - iy;
    t.wz1 = iz - (float)t.z0, t.wz0 = (float)(t.z0 + 1) - iz;
    return t;
}
__device__ __forceinline__ bool tri_inb(const TriG &t, int c, int D, int H, int W, int &vox, float &w) {
    const int x = t.x0 + (c & 1), y = t.y0 + ((c >> 1) & 1), z = t.z0 + (c >> 2);
    w = ((c & 1) ? t.wx1 : t.wx0) * (((c >> 1) & 1) ? t.wy1 : t.wy0) * ((c >> 2) ? t.wz1 : t.wz0);
    vox = (z * H + y) * W + x;
    return x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D;
}
// ---- the same two lookups without branches (round 6; the forward's sampler in both sweeps, and the primitive-centric
// backward's) ---------------------------------------------------------------------------------------------------------
// tri_general + tri_inb guard every corner with its own bounds test: eight exec-mask regions per lookup, each with its own
// load and its own wait (the use sits inside the region) -- eight dependent round trips per lookup where the plain sampler
// has one.  Here every corner is READ: along each axis the two cells are clamped into the slab separately,
//   ca = clamp(f, 0, T-1), cb = clamp(f + 1, 0, T-1)   (f = floor(i)),
// and a corner that the reference's zero padding leaves out gets the WEIGHT zero instead (utils.h:459-498: a term that is
// not added = a term times zero, for finite cells).  A corner outside the slab along one axis reads the cell of the corner
// INSIDE along that axis (ca == cb there), so a non-finite cell poisons the sample only if the reference reads that cell
// too; a sample with an axis entirely outside has no corner the reference reads and is forced to zero.  (What differs: a
// +-Inf cell read through a zero weight gives NaN here where the reference's sum stays +-Inf -- both non-finite.)
struct AxisZ {
    float ca, cb;  // the two cells read along this axis (small integers held as floats)
    float wa, wb;  // their weights: (x0 + 1) - i and i - x0 where the reference's corner is in bounds, else 0
    float sa, sb;  // d(wa)/di, d(wb)/di: -1 and +1 where the corner is in bounds, else 0 (backward only)
    bool live;     // some corner along this axis is in bounds
};
__device__ __forceinline__ AxisZ axis_zero_pad(float yn, float tm1 /* T - 1 */) {
#pragma clang fp contract(off)
    const float i = fmaxf(-100.f, fminf(100.f, (yn + 1.f) * 0.5f)) * tm1;  // utils.h:416-418
    const float f = floorf(i);
    AxisZ a;
    a.ca = fminf(fmaxf(f, 0.f), tm1);
    a.cb = fminf(fmaxf(f + 1.f, 0.f), tm1);
    const bool ina = f >= 0.f && f <= tm1, inb = f >= -1.f && f < tm1;
    a.wa = ina ? (f + 1.f) - i : 0.f;
    a.wb = inb ? i - f : 0.f;
    a.sa = ina ? -1.f : 0.f;
    a.sb = inb ? 1.f : 0.f;
    a.live = f >= -1.f && f <= tm1;
    return a;
}
struct TriZ {
    uint32_t c[8];           // cell indices of the corners 000, 001 (x + 1), 010 (y + 1), ..., 111: all inside the grid
    v2f W00, W01, W10, W11;  // the natural weight pairs W_zy = (w_zy0, w_zy1), as in TriF
    v2f wyzA, wyzB;          // (w_y0 w_z0, w_y1 w_z0), (w_y0 w_z1, w_y1 w_z1)
    AxisZ ax, ay, az;
    bool live;
};
__device__ __forceinline__ TriZ tri_zero_pad(f3 y, int D, int H, int W) {
#pragma clang fp contract(off)
    TriZ t;
    t.ax = axis_zero_pad(y.x, (float)(W - 1)), t.ay = axis_zero_pad(y.y, (float)(H - 1)), t.az = axis_zero_pad(y.z, (float)(D - 1));
    t.live = t.ax.live && t.ay.live && t.az.live;
    // (z * H + y) * W + x in float: integers below 2^24 (the host checks the grid sizes), exact
    const float fH = (float)H, fW = (float)W;
    const float r00 = __builtin_fmaf(t.az.ca, fH, t.ay.ca), r01 = __builtin_fmaf(t.az.ca, fH, t.ay.cb);
    const float r10 = __builtin_fmaf(t.az.cb, fH, t.ay.ca), r11 = __builtin_fmaf(t.az.cb, fH, t.ay.cb);
    t.c[0] = (uint32_t)__builtin_fmaf(r00, fW, t.ax.ca), t.c[1] = (uint32_t)__builtin_fmaf(r00, fW, t.ax.cb);
    t.c[2] = (uint32_t)__builtin_fmaf(r01, fW, t.ax.ca), t.c[3] = (uint32_t)__builtin_fmaf(r01, fW, t.ax.cb);
    t.c[4] = (uint32_t)__builtin_fmaf(r10, fW, t.ax.ca), t.c[5] = (uint32_t)__builtin_fmaf(r10, fW, t.ax.cb);
    t.c[6] = (uint32_t)__builtin_fmaf(r11, fW, t.ax.ca), t.c[7] = (uint32_t)__builtin_fmaf(r11, fW, t.ax.cb);
    const v2f wxp{t.ax.wa, t.ax.wb}, wyp{t.ay.wa, t.ay.wb}, wzp{t.az.wa, t.az.wb};
    t.wyzA = pk_mul_lo(wyp, wzp), t.wyzB = pk_mul_hi(wyp, wzp);  // (wyz00, wyz10), (wyz01, wyz11)
    t.W00 = pk_mul_lo(wxp, t.wyzA), t.W01 = pk_mul_hi(wxp, t.wyzA);
    t.W10 = pk_mul_lo(wxp, t.wyzB), t.W11 = pk_mul_hi(wxp, t.wyzB);
    return t;
}
// d(zero-padded trilinear form)/d(index position) for channel-dotted corner values d_zyx (utils.h:592-642: over the corners in
// bounds, -+1 along the axis times the other two weights), as a tree: the four x edges give d/dx and the edge values, their y
// combinations d/dy, the last one d/dz
__device__ __forceinline__ f3 posgrad_zero_pad(const TriZ &t, float d000, float d001, float d010, float d011, float d100,
                                               float d101, float d110, float d111) {
    const float wxa = t.ax.wa, wxb = t.ax.wb, sxa = t.ax.sa, sxb = t.ax.sb;
    const float e00 = fmaf(wxb, d001, wxa * d000), e01 = fmaf(wxb, d011, wxa * d010);  // (z0,y0) (z0,y1)
    const float e10 = fmaf(wxb, d101, wxa * d100), e11 = fmaf(wxb, d111, wxa * d110);  // (z1,y0) (z1,y1)
    const float s00 = fmaf(sxb, d001, sxa * d000), s01 = fmaf(sxb, d011, sxa * d010);
    const float s10 = fmaf(sxb, d101, sxa * d100), s11 = fmaf(sxb, d111, sxa * d110);
    f3 g;
    g.x = fmaf(t.wyzB.y, s11, fmaf(t.wyzB.x, s10, fmaf(t.wyzA.y, s01, t.wyzA.x * s00)));
    const float f0 = fmaf(t.ay.wb, e01, t.ay.wa * e00), f1 = fmaf(t.ay.wb, e11, t.ay.wa * e10);
    const float t0 = fmaf(t.ay.sb, e01, t.ay.sa * e00), t1 = fmaf(t.ay.sb, e11, t.ay.sa * e10);
    g.y = fmaf(t.az.wb, t1, t.az.wa * t0);
    g.z = fmaf(t.az.sb, f1, t.az.sa * f0);
    return g;
}
struct __attribute__((packed, aligned(4))) Node3 {  // one node of a warp grid: 12 bytes, ONE dwordx3 gather
    float x, y, z;
};
// primsampler.h:48-63 with dowarp: fade from y, template sampled at y1 = warp(y) with zero padding, (r, g, b, alpha * fade).
// y strictly inside (-1,1)^3: every corner of the warp lookup is in bounds (the general form then gives the clamped one).
// Both forward sweeps call this one function (explicit fused operations: the same bits at both call sites).
template <bool FADE8>
__device__ __forceinline__ float4 sample_warped(const float *__restrict__ Wk, const float *__restrict__ Tk, f3 y, int WD,
                                                int WH, int WW, int TD, int TH, int TW, float fadescale, float fadeexp) {
#pragma clang fp contract(off)
    const float fade = fade_pinned<FADE8>(y, fadescale, fadeexp);
    const TriZ tw = tri_zero_pad(y, WD, WH, WW);
    const Node3 *Wn = reinterpret_cast<const Node3 *>(Wk);
    const Node3 n0 = Wn[tw.c[0]], n1 = Wn[tw.c[1]], n2 = Wn[tw.c[2]], n3 = Wn[tw.c[3]];
    const Node3 n4 = Wn[tw.c[4]], n5 = Wn[tw.c[5]], n6 = Wn[tw.c[6]], n7 = Wn[tw.c[7]];
    f3 y1;
#define MVP_WSUM(M_)                                                                                              \
    __builtin_fmaf(n7.M_, tw.W11.y, __builtin_fmaf(n6.M_, tw.W11.x, __builtin_fmaf(n5.M_, tw.W10.y,               \
    __builtin_fmaf(n4.M_, tw.W10.x, __builtin_fmaf(n3.M_, tw.W01.y, __builtin_fmaf(n2.M_, tw.W01.x,               \
    __builtin_fmaf(n1.M_, tw.W00.y, n0.M_ * tw.W00.x)))))))
    y1.x = MVP_WSUM(x), y1.y = MVP_WSUM(y), y1.z = MVP_WSUM(z);
#undef MVP_WSUM
    const TriZ tt = tri_zero_pad(y1, TD, TH, TW);
    const float4 *T4 = reinterpret_cast<const float4 *>(Tk);
    const float4 c000 = T4[tt.c[0]], c001 = T4[tt.c[1]], c010 = T4[tt.c[2]], c011 = T4[tt.c[3]];
    const float4 c100 = T4[tt.c[4]], c101 = T4[tt.c[5]], c110 = T4[tt.c[6]], c111 = T4[tt.c[7]];
    TriF tf;
    tf.off = 0u, tf.W00 = tt.W00, tf.W01 = tt.W01, tf.W10 = tt.W10, tf.W11 = tt.W11;
    float4 v = tri_interp(tf, c000, c001, c010, c011, c100, c101, c110, c111);
    if (!tt.live) v = make_float4(0.f, 0.f, 0.f, 0.f);
    v.w = v.w * fade;
    return v;
}
template <bool FADE8>
__device__ __forceinline__ float fade_of(f3 y, float fadescale, float fadeexp) {
    if (FADE8) {
        const f3 y2 = y * y, y4 = y2 * y2;
        return fast_exp(-fadescale * (y4.x * y4.x + y4.y * y4.y + y4.z * y4.z));
    }
    return fast_exp(-fadescale *
                    (fast_pow(fabsf(y.x), fadeexp) + fast_pow(fabsf(y.y), fadeexp) + fast_pow(fabsf(y.z), fadeexp)));
}
// d(trilinear)/d(position) in index units for channel-dotted corner values `dotc` (utils.h:592-642): returns
// (sum +-wy*wz*dot, sum +-wx*wz*dot, sum +-wx*wy*dot) over the in-bounds corners
__device__ __forceinline__ void tri_posgrad_acc(const TriG &t, int c, float dot, f3 &g) {
    const float wx = (c & 1) ? t.wx1 : t.wx0, wy = ((c >> 1) & 1) ? t.wy1 : t.wy0, wz = (c >> 2) ? t.wz1 : t.wz0;
    g.x += ((c & 1) ? 1.f : -1.f) * wy * wz * dot;
    g.y += (((c >> 1) & 1) ? 1.f : -1.f) * wx * wz * dot;
    g.z += ((c >> 2) ? 1.f : -1.f) * wx * wy * dot;
}

// Lattice steps s (t_s = tmin + s*dt) of one ray that can fall strictly inside a box whose slab interval is
// [tn, tf] (utils.h:747-753), clipped to the ray's [tmin, tmax + 1e-5).  The strict inside test on the evaluated
// position decides membership exactly as in the reference; this range only has to contain every step that test
// can accept, so it is the analytic range widened by a slack that covers fp32 rounding of t and of the slab test
// (a few 1e-7 * |t| / dt steps) -- not by whole steps, which would waste one third of the march iterations.
__device__ __forceinline__ bool lane_step_range(float tn, float tf, float tmin, float tmax, float dt, int &lo,
                                                int &hi) {
    const float ta = fmaxf(tn, tmin), tb = fminf(tf, tmax + 1e-5f);
    if (!(tn <= tf) || !(ta <= tb)) return false;
    const float idt = fast_rcp(dt);  // the slack below is ~1e5 times the rounding this can add
    const float slack = 0.02f + 2.0e-6f * fmaxf(fmaxf(fabsf(ta), fabsf(tb)), 1.f) * idt;
    const float flo = ceilf((ta - tmin) * idt - slack), fhi = floorf((tb - tmin) * idt + slack);
    lo = (int)fminf(fmaxf(flo, 0.f), 1.0e9f);
    hi = (int)fminf(fmaxf(fhi, 0.f), 1.0e9f);
    return lo <= hi;
}

constexpr int kPrimGranule = 128;   // primitives per granule when F XCDs share an image (block -> primitive mapping)
// block slots of one XCD's share of an image's K primitives: whole granules, ceil(granules / F) of them
__host__ __device__ constexpr int prim_band_slots(int K, int F) {
    return (((K + kPrimGranule - 1) / kPrimGranule + F - 1) / F) * kPrimGranule;
}

// ---- packet -> (image, tile).  Block b runs on XCD b % 8 (MI355X_MICROARCH "Workgroup dispatch"), so the block
// index decides which XCD renders what, statically.  An image is a sequence of strips (kStripRows packet rows,
// walked column by column: the packets that share a primitive's slab -- it spans ~2 x 2 packets at C2 -- start a few
// blocks apart instead of a row apart).  F XCDs share an image by taking its strips cyclically, 8 / F images are in
// flight at a time:
//   * the first N - N % 8 images: F = 1, XCD x renders images x, x + 8, ... whole;
//   * the other R = N % 8 images (all of them when N < 8): F = 2 for R = 4, 4 for R = 2, else 8.
// Why: the first version gave XCD x the x-th horizontal BAND of every image -- the top and bottom bands of a head
// shot are background, so two XCDs idled while the two middle ones carried the kernel (same total wave-cycles, 30 %
// longer wall time).  C2 forward 7.51 ms (bands, row-major) -> 7.04 (bands, strips) -> 5.30 (whole images); C3 / C4
// (N = 4) 0.81 / 1.12 -> 0.70 / 0.93 with two half-image bands per image -> see DESIGN.md 3.3 for the cyclic form.
__host__ __device__ inline bool packet_of_block(const MarchParams &p, int b, int &n, int &tidx) {
    const int T8 = 8 * p.chunk, blocks_whole = p.images_whole * T8;
    int j, F, band;  // packet slot inside this XCD's share of the image, XCDs per image, which of them
    if (b < blocks_whole) {
        const int xcd = b & 7, i = b >> 3, q = i / T8;
        n = q * 8 + xcd, j = i - q * T8, F = 1, band = 0;
    } else {
        const int bb = b - blocks_whole, xcd = bb & 7, i = bb >> 3, q = i / p.band_chunk;
        F = p.band_split, band = xcd % F;
        n = p.images_whole + q * (8 / F) + xcd / F, j = i - q * p.band_chunk;
        if (n >= p.N) return false;
    }
    const int S = kStripRows * p.tiles_x;          // packet slots per strip
    int m = j / S;                                  // this XCD's m-th strip of the image
    if (F > 1) {
        // XCDs that SHARE an image walk their strips in two interleaved runs -- 0, h, 1, h + 1, ... with h = half of the strips --
        // instead of top-down: the top run gets heavier while the bottom run gets lighter (a head shot: background, head,
        // background), so packets that only traverse and packets that sweep slabs are in flight together all the way instead
        // of a light, a heavy and a light phase.  With 4 images on the chip the whole kernel is ~3 generations of packets and
        // the phases do not average out (round 6: C3 forward 0.712 -> 0.684 ms, C4 0.950 -> 0.892; centre-out, i.e. heavy
        // first, LOSES 7 % / 2 %).  Whole images (F = 1: tens of generations per XCD) keep the top-down order: the same
        // permutation costs 1 % there, two runs of slabs competing for the 4 MB L2 (profiles/r06_fwd_strip_order.txt).
        const int NS = (p.tiles_y + kStripRows - 1) / kStripRows, M = (NS + F - 1) / F, half = (M + 1) / 2;
        if (m >= M) return false;
        m = (m & 1) ? half + (m >> 1) : (m >> 1);
    }
    const int strip = m * F + band, jj = j % S;  // the strip of the image, the slot inside it
    const int row0 = strip * kStripRows;
    const int rows = p.tiles_y - row0 < kStripRows ? p.tiles_y - row0 : kStripRows;
    if (rows <= 0 || jj >= rows * p.tiles_x) return false;  // (a ragged last strip leaves some slots empty)
    tidx = (row0 + jj % rows) * p.tiles_x + jj / rows;
    return true;
}

// Block -> (image, primitive) of the primitive-centric backward (block b runs on XCD b % 8): XCD x owns ALL primitives of
// images x, x + 8, ... of the first N - N % 8 images -- an image's ray records, which ~7 of its primitives re-read, then
// live in one L2 instead of eight; the remaining R images are split over F = band_split XCDs each, 8 / F images at a
// time (see packet_of_block), in granules of kPrimGranule primitives (neighbours on the shell share rays) dealt
// cyclically; contiguous ranges of k were 4.5 % slower at C4, equal at C3.
__host__ __device__ inline bool prim_of_block(const MarchParams &p, int b, int &n, int &k) {
    const int K = p.K, blocks_whole = p.images_whole * K;
    if (b < blocks_whole) {
        const int xcd = b & 7, i = b >> 3, q = i / K;
        n = q * 8 + xcd, k = i - q * K;
        return true;
    }
    const int bb = b - blocks_whole, xcd = bb & 7, i = bb >> 3;
    const int F = p.band_split, slots = prim_band_slots(K, F), q = i / slots, il = i - q * slots;
    n = p.images_whole + q * (8 / F) + xcd / F;
    k = ((il / kPrimGranule) * F + xcd % F) * kPrimGranule + il % kPrimGranule;
    return n < p.N && k < K;
}

// ---- host side shared by the forward and backward entry points (march_host.hip) ---------------------------------------
// grid geometry of the march kernels from (N, H, W, K): fills the fields packet_of_block / prim_of_block read
__attribute__((visibility("hidden"))) int setup_block_map(MarchParams &p);
// blocks of the primitive-centric backward's grid
__attribute__((visibility("hidden"))) long long prim_grid_blocks(const MarchParams &p);
// argument checks of both directions; returns MVP_OK, an error, or 1 when there are no rays (nothing to do)
__attribute__((visibility("hidden"))) int march_common_checks(bool bwd, MarchParams &p);

}  // namespace mvp
