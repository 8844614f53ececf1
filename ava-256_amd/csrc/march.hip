// march.hip -- Mixture-of-Volumetric-Primitives raymarch, forward and backward, hand-written for gfx950 (CDNA4).
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
#include <stdlib.h>

#include <type_traits>
#include "mvp_device.h"
#include "mvp_host.h"

namespace mvp {

constexpr int kTile = 8;          // 8x8 pixels per wave
constexpr int kMaxList = 512;     // reference hit-list cap (mvpraymarch_kernel.cu:101, utils.h:779)
#ifndef MVP_REC_SLOTS
#define MVP_REC_SLOTS 64
#endif
constexpr int kRecSlots = MVP_REC_SLOTS;  // SRT records staged in LDS (first 64 candidates); beyond: scalar global loads
constexpr int kStartDepth = 10;   // the BFS tests every node of this depth first (implicit frontier, <= 1024 nodes)
constexpr int kNoSlot = 255;
// Lane-independent forward sweep (see march_packet): per-ray crossing table in LDS, kFastCross rows of 64 lanes.
// Row index 31 is the null link, so a ray can hold at most min(kFastCross, 31) crossings; packets beyond any of the
// limits below are marched by the slot-synchronous sweep instead (same results, slower).  kFastSlots records (64 B each)
// + kFastCross rows (256 B each) share the 8 KB the slot-synchronous layout needs: 40 + 22 keeps 5 waves per SIMD and
// measured best over C2/C3/C4 (64 + 24 at 10 KB, 48 + 20, 56 + 18 and 64 + 16 at 8 KB were within 3 %).
#ifndef MVP_FAST_CROSS
#define MVP_FAST_CROSS 22
#endif
constexpr int kFastCross = MVP_FAST_CROSS;
constexpr int kFastMaxCross = kFastCross < 31 ? kFastCross : 31;
#ifndef MVP_FAST_RECS
#define MVP_FAST_RECS 40
#endif
constexpr int kFastSlots = MVP_FAST_RECS;  // list slots (6-bit field; every record of the fast path is LDS-resident) =
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
    uint2 *pl_list;       // [N*K, pl_cap]: {(packet << 9) | list slot, lo | hi << 16}
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
#ifndef MVP_STRIP_ROWS
#define MVP_STRIP_ROWS 3  // packet rows per dispatch strip (march_packet: packet -> (image, tile))
#endif

// Streaming traffic is marked non-temporal so that it does not push re-used lines out of the L2: in the backward a
// primitive's slab is read once and its gradient written once per launch (5.4 GB at C2) while the ray records the same
// workgroups gather are re-read by the ~7 primitives a ray crosses; in the forward the rays are read once and the
// hand-off records (raysat, rayaux) are not read again before the backward, while slab lines are shared by neighbouring
// packets.
#ifdef MVP_NO_STREAM_HINTS
#define MVP_STREAM_LOAD(P_) (*(P_))
#define MVP_STREAM_STORE(P_, V_) (*(P_) = (V_))
#define MVP_STREAM_LOADF(P_) (*(P_))
#define MVP_STREAM_STOREF(P_, V_) (*(P_) = (V_))
#else
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
#endif

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
    const float ird = 1.0f / d;
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

// ---- warp-field path (algo 1: PrimSamplerTW<true>, primsampler.h:53-58,82-88) -------------------------------------
// The warped coordinate y1 may leave (-1,1)^3, so the template lookup needs the reference's general form: normalised
// coordinate clamped to +-100, floor, zero padding through per-corner bounds tests (utils.h:414-498).
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
// y1 = trilinear 3-channel lookup of the warp grid at y0 (y0 strictly inside: every corner is in bounds after the
// base-corner clamp, same value as the zero-padded form)
__device__ __forceinline__ f3 warp_lookup(const float *__restrict__ Wk, f3 y0, int D, int H, int W) {
    TriG t = tri_general(y0, D, H, W);
    f3 r = mk3(0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        int vox;
        float w;
        if (tri_inb(t, c, D, H, W, vox, w)) {
            const float *q = Wk + (size_t)vox * 3;
            r.x += q[0] * w, r.y += q[1] * w, r.z += q[2] * w;
        }
    }
    return r;
}
__device__ __forceinline__ float4 tplate_lookup_general(const float *__restrict__ Tk, f3 y1, int D, int H, int W) {
    TriG t = tri_general(y1, D, H, W);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        int vox;
        float w;
        if (tri_inb(t, c, D, H, W, vox, w)) {
            const float4 q = *reinterpret_cast<const float4 *>(Tk + (size_t)vox * 4);
            v.x += q.x * w, v.y += q.y * w, v.z += q.z * w, v.w += q.w * w;
        }
    }
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
// index decides which XCD renders what, statically.  An image is a sequence of strips (MVP_STRIP_ROWS packet rows,
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
    const int S = MVP_STRIP_ROWS * p.tiles_x;          // packet slots per strip
    const int strip = (j / S) * F + band, jj = j % S;  // the strip of the image, the slot inside it
    const int row0 = strip * MVP_STRIP_ROWS;
    const int rows = p.tiles_y - row0 < MVP_STRIP_ROWS ? p.tiles_y - row0 : MVP_STRIP_ROWS;
    if (rows <= 0 || jj >= rows * p.tiles_x) return false;  // (a ragged last strip leaves some slots empty)
#ifndef MVP_NO_STRIP_ORDER
    tidx = (row0 + jj % rows) * p.tiles_x + jj / rows;
#else
    tidx = row0 * p.tiles_x + jj;
#endif
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

// LDS ordering point inside a packet.  One wave = one workgroup: a workgroup barrier (which is free there).  Timing variant
// MVP_FWD_QUAD > 1 (several independent packets = waves per workgroup, so that neighbouring packets share a CU's L1): a fence
// only -- every wave works in its own LDS block and a wave's LDS operations complete in order; an s_barrier would tie
// packets with different trip counts together.
#ifndef MVP_FWD_QUAD
#define MVP_FWD_QUAD 1
#endif
__device__ __forceinline__ void packet_sync() {
#if MVP_FWD_QUAD > 1
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
#else
    __syncthreads();
#endif
}

template <bool BWD, bool FADE8, bool WARP, int TS>
__device__ __forceinline__ void march_packet(const MarchParams &p, const int b, int *s_a, int *s_b, float4 *s_rec,
                                             uint32_t *s_tab, const bool emit_all) {
    constexpr bool FAST = !BWD && !WARP;  // the lane-independent sweep exists for the plain forward only
    const int lane = lane_id();
    const unsigned long long lt = lanemask_lt(lane);

    int n, tidx;
    if (!packet_of_block(p, b, n, tidx)) return;
    if (BWD && !emit_all) {
        // ray-centric backward for a FEW primitives: only the packets on their lists do anything -- marked by the forward
        // (the packets it could not append) and by the primitive-centric kernel (the ones it could)
        const uint32_t w_ = p.pl_count[(size_t)p.N * p.K + 3 + (size_t)n * p.tiles_x * p.tiles_y + tidx];
        if ((w_ & (kPacketFwdOverflow | kPacketBwdWanted)) == 0u) return;
    }
    const int ty = tidx / p.tiles_x, tx = tidx - ty * p.tiles_x;
    const int px = tx * kTile + (lane & 7), py = ty * kTile + (lane >> 3);
    const bool inimg = px < p.W && py < p.H;
    const size_t r = ((size_t)n * p.H + (inimg ? py : 0)) * p.W + (inimg ? px : 0);

    const int K = p.K, NN = 2 * K - 1;
    const float dt = p.stepsize;
    const float *pp = p.primpos + (size_t)n * K * 3;
    const float *pr = p.primrot + (size_t)n * K * 9;
    const float *ps = p.primscale + (size_t)n * K * 3;
    const float *A = p.nodeaabb + (size_t)n * NN * 6;

    f3 o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 1.f);
    float tmin = INFINITY, tmax = -INFINITY;
    if (inimg) {
        if (!BWD && p.campos != nullptr) {  // (wave-uniform) rays from the camera: no ray tensors are read
            float fpx = (float)px, fpy = (float)py;
            if (p.pixelcoords) {
                const float2 pc = reinterpret_cast<const float2 *>(p.pixelcoords)[r];
                fpx = pc.x, fpy = pc.y;
            }
            const float *cp = p.campos + (size_t)n * 3, *fo = p.focal + (size_t)n * 2, *pc2 = p.princpt + (size_t)n * 2;
            const CamRay c = ray_from_camera(mk3(cload(cp), cload(cp + 1), cload(cp + 2)), p.camrot + (size_t)n * 9,
                                             cload(fo), cload(fo + 1), cload(pc2), cload(pc2 + 1), fpx, fpy, p.volradius);
            o = c.o, d = c.d, tmin = c.tmin, tmax = c.tmax;
            if (p.raydir_out) {  // (wave-uniform) the backward's ray tensors, written by the packet that owns the pixel
                float *po = p.raypos_out + r * 3, *pd = p.raydir_out + r * 3;
                po[0] = o.x, po[1] = o.y, po[2] = o.z;
                pd[0] = d.x, pd[1] = d.y, pd[2] = d.z;
                reinterpret_cast<float2 *>(p.tminmax_out)[r] = make_float2(tmin, tmax);
            }
        } else {
            const float *op = p.raypos + r * 3, *dp = p.raydir + r * 3, *tp = p.tminmax + r * 2;
            o = mk3(MVP_STREAM_LOADF(op), MVP_STREAM_LOADF(op + 1), MVP_STREAM_LOADF(op + 2));
            d = mk3(MVP_STREAM_LOADF(dp), MVP_STREAM_LOADF(dp + 1), MVP_STREAM_LOADF(dp + 2));
            tmin = MVP_STREAM_LOADF(tp);
            tmax = MVP_STREAM_LOADF(tp + 1);
        }
    }
    // a ray can only take a sample at t in [tmin, tmax + 1e-5) (subset_kernel.h:63-64,84)
    const bool active = inimg && (tmin < tmax + 1e-5f);

    float4 rgba = make_float4(0.f, 0.f, 0.f, 0.f);
    f3 raysat = mk3(-1.f, -1.f, -1.f);
    uint32_t satkey = kNoSat;  // (step << 9) | list slot of the saturating sample
    float wbefore = 0.f;       // alpha just before it
    // A NaN opacity sample (a diverged decoder).  primaccum.h:66-67: fminf(NaN, 1) = 1, so the forward fills alpha up to 1 and
    // saturates at the NEXT sample -- while the reference's backward recomputes the prefix, gets NaN, never sees "saturated"
    // and gives every later sample of the ray the unsaturated weight (primaccum.h:86-95).  The forward's record cannot
    // express that; the packet raises the global flag instead and the ray-centric kernel, which recomputes the prefix the
    // same way, owns this backward (slow, exact; the loop zeroes such gradients anyway, ddp-train.py:436-439).
    bool nanw = false;
    int nh = 0;            // final list length (wave-uniform)
    int ncand = 0;
    bool fast = false;     // wave-uniform: this packet is marched by the lane-independent sweep
    int kk0 = 0, kk1 = 0;  // candidates `lane` and `lane + 64` (lane-independent mode)

    if (__ballot(active) != 0ull) {
        // ---------------- packet bounds (6-step butterflies, once per packet) ----------------
        PacketBounds pb;
        // (the first ACTIVE lane's origin, and whether every active lane has it: one ballot)
        const int fl = __ffsll((long long)__ballot(active)) - 1;
        const f3 of = mk3(rl_f(o.x, fl), rl_f(o.y, fl), rl_f(o.z, fl));
        const bool same_o = __ballot(active && (o.x != of.x || o.y != of.y || o.z != of.z)) == 0ull;
        pb.ax = axis_bounds(active, o.x, d.x, same_o, of.x);
        pb.ay = axis_bounds(active, o.y, d.y, same_o, of.y);
        pb.az = axis_bounds(active, o.z, d.z, same_o, of.z);
        pb.tlo = uni(wave_min(active ? tmin : INFINITY));
        pb.thi = uni(wave_max(active ? tmax + 1e-5f : -INFINITY));

        // ---------------- breadth-first frontier expansion, lanes over nodes ----------------
        // The fixed-order heap is a poor BVH near the root (a depth-d node is K/2^d CONSECUTIVE primitives: a ring
        // of the shell, a row of the UV grid), so its upper levels cull nothing.  The packet therefore tests the
        // root once (most empty packets leave here), then ALL nodes of depth ds (up to 1024, implicit: nothing is
        // stored) in 64-lane rounds of independent loads, and only then walks the remaining <= 4 levels with an
        // explicit, compacted frontier.
        const int dmax = 31 - __clz(NN);  // depth of the deepest node; depth(i) = floor(log2(i+1))
        // leaves sit at depth dmax or dmax-1: start no deeper than dmax-1 so that none is skipped
        const int ds = max(0, min(dmax - 1, kStartDepth));
        int *cur = s_a, *nxt = s_b;
        const int first = (1 << ds) - 1;
        int ncur = min(1 << ds, NN - first);
        bool frontier_ovf = false;
        {
            const float2 *ap = reinterpret_cast<const float2 *>(A);  // root AABB, wave-uniform
            const float2 a0 = ap[0], a1 = ap[1], a2 = ap[2];
            if (!packet_hits_box(pb, a0.x, a0.y, a1.x, a1.y, a2.x, a2.y)) ncur = 0;
            if (MVP_DEBUG_STAGE(p) == 11) ncur = 0;
        }
        // Coarse pre-cull of the implicit level: its nodes are grouped 32 per ancestor 5 levels up (<= 32 ancestors,
        // one lane each); groups whose ancestor fails the packet test are skipped without touching their boxes.
        unsigned anc_pass = 0xffffffffu;
        if (ncur > 0 && ds >= 5) {
            const int nanc = 1 << (ds - 5);
            bool ok = false;
            if (lane < nanc) {
                const float2 *ap = reinterpret_cast<const float2 *>(A + (size_t)(nanc - 1 + lane) * 6);
                const float2 a0 = ap[0], a1 = ap[1], a2 = ap[2];
                ok = packet_hits_box(pb, a0.x, a0.y, a1.x, a1.y, a2.x, a2.y);
            }
            anc_pass = (unsigned)__ballot(ok);
            if (anc_pass == 0u || MVP_DEBUG_STAGE(p) == 12) ncur = 0;
        }
        for (int dep = ds; ncur > 0; ++dep) {
            int nnext = 0;
            for (int base = 0; base < ncur; base += kWave) {
                const int idx = base + lane;
                if (dep == ds && ((anc_pass >> (base >> 5)) & 3u) == 0u) continue;  // both ancestor groups culled
                const bool have = idx < ncur && (dep != ds || ((anc_pass >> (idx >> 5)) & 1u) != 0u);
                const int e = !have ? 0 : (dep == ds ? first + idx : cur[idx]);
                const bool tested_leaf = e < 0;  // ~node: a leaf that already passed, carried to keep order
                const int g = tested_leaf ? ~e : e;
                bool pass = have && tested_leaf;
                if (have && !tested_leaf) {
                    const float2 *ap = reinterpret_cast<const float2 *>(A + (size_t)g * 6);  // 24 B nodes: 8-B aligned
                    const float2 a0 = ap[0], a1 = ap[1], a2 = ap[2];
                    pass = packet_hits_box(pb, a0.x, a0.y, a1.x, a1.y, a2.x, a2.y);
                }
                const bool isleaf = g >= K - 1;
                const bool e1 = pass, e2 = pass && !isleaf;
                const unsigned long long m1 = __ballot(e1), m2 = __ballot(e2);
                const int pos = nnext + __popcll(m1 & lt) + __popcll(m2 & lt);
                if (e1 && pos < kMaxList) nxt[pos] = isleaf ? ~g : 2 * g + 1;
                if (e2 && pos + 1 < kMaxList) nxt[pos + 1] = 2 * g + 2;
                nnext += __popcll(m1) + __popcll(m2);
            }
            if (nnext > kMaxList || MVP_DEBUG_FORCE_DFS(p)) {
                frontier_ovf = true;
                break;
            }
            packet_sync();
            int *t = cur;
            cur = nxt;
            nxt = t;
            ncur = nnext;
            if (MVP_DEBUG_STAGE(p) == 13) ncur = 0;
            if (dep >= dmax) break;
        }
        ncand = ncur;  // entries of `cur` are ~node of tested leaves, in DFS (left-to-right) order
        if (frontier_ovf) {
            // ---- exact fallback: the reference's own traversal (utils.h:733-814): wave-uniform DFS with an explicit
            //      stack, every lane testing ITS ray against both children (utils.h:679-685), decisions OR-ed over
            //      the packet.  Slow (one dependent round trip per node) but capacity-free; only heavy scenes get here.
            if (p.diag && lane == 0) atomicAdd(p.diag + MVP_DIAG_FRONTIER_OVERFLOW, 1u);
            packet_sync();
            int *stack = s_a;  // wave-uniform contents
            int *cand = s_b;
            const f3 irdl = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            int sp = 0, node = 0;
            ncand = 0;
            while (node != -1) {
                if (node >= K - 1) {
                    // leaf: exact ray/box test as in utils.h:744-755 so that only real hits count toward the
                    // 512-entry capacity (the pass below repeats it to get the step ranges)
                    const Rec q = rec_from_global(pp, pr, ps, node - (K - 1));
                    const f3 r0 = rot_rows(q, o - q.pos) * q.scale, rd = rot_rows(q, d) * q.scale;
                    const f3 ird = mk3(fast_rcp(rd.x), fast_rcp(rd.y), fast_rcp(rd.z));
                    const f3 t0 = mk3((-1.f - r0.x) * ird.x, (-1.f - r0.y) * ird.y, (-1.f - r0.z) * ird.z);
                    const f3 t1 = mk3((1.f - r0.x) * ird.x, (1.f - r0.y) * ird.y, (1.f - r0.z) * ird.z);
                    const bool hit = active && max3f(fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z)) <=
                                                   min3f(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y), fmaxf(t0.z, t1.z));
                    if (__ballot(hit) != 0ull) {
                        if (ncand < kMaxList) {
                            if (lane == 0) cand[ncand] = ~node;
                            ++ncand;
                        } else if (p.diag && lane == 0) {  // the reference drops these too (utils.h:779)
                            atomicAdd(p.diag + MVP_DIAG_LIST_OVERFLOW, 1u);
                        }
                    }
                    node = sp > 0 ? uni(stack[--sp]) : -1;
                } else {
                    const int cl = 2 * node + 1;
                    const float *bx = A + (size_t)cl * 6;  // both children: 12 consecutive floats, wave-uniform address
                    bool hl, hr;
                    {
                        const f3 t0 = mk3((bx[0] - o.x) * irdl.x, (bx[1] - o.y) * irdl.y, (bx[2] - o.z) * irdl.z);
                        const f3 t1 = mk3((bx[3] - o.x) * irdl.x, (bx[4] - o.y) * irdl.y, (bx[5] - o.z) * irdl.z);
                        hl = max3f(fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z)) <=
                             min3f(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y), fmaxf(t0.z, t1.z));
                        const f3 u0 = mk3((bx[6] - o.x) * irdl.x, (bx[7] - o.y) * irdl.y, (bx[8] - o.z) * irdl.z);
                        const f3 u1 = mk3((bx[9] - o.x) * irdl.x, (bx[10] - o.y) * irdl.y, (bx[11] - o.z) * irdl.z);
                        hr = max3f(fminf(u0.x, u1.x), fminf(u0.y, u1.y), fminf(u0.z, u1.z)) <=
                             min3f(fmaxf(u0.x, u1.x), fmaxf(u0.y, u1.y), fmaxf(u0.z, u1.z));
                    }
                    const bool tl = __ballot(active && hl) != 0ull, tr = __ballot(active && hr) != 0ull;
                    if (!tl && !tr) {
                        node = sp > 0 ? uni(stack[--sp]) : -1;
                    } else {
                        node = tl ? cl : cl + 1;
                        if (tl && tr) {  // depth <= 31 < kMaxList entries
                            if (lane == 0) stack[sp] = cl + 1;
                            ++sp;
                        }
                    }
                    packet_sync();
                }
            }
            packet_sync();
            cur = cand;
        }

        // ---- candidates leave the frontier buffers (cur may be either one) ----
        // Lane-independent mode (FAST, <= kFastCand candidates): they stay in two registers per lane, because the
        // frontier region is about to become the per-ray crossing table.  Otherwise: s_b, as primitive indices.
        fast = FAST && ncand > 0 && ncand <= kFastCand && !MVP_DEBUG_SLOT_SWEEP(p);
        if (ncand > 0) {
            if (fast) {
                kk0 = lane < ncand ? (~cur[lane]) - (K - 1) : 0;
                kk1 = lane + kWave < ncand ? (~cur[lane + kWave]) - (K - 1) : 0;
                packet_sync();
            } else {
                int kk[kMaxList / kWave];
#pragma unroll
                for (int c = 0; c < kMaxList / kWave; ++c) {
                    const int idx = c * kWave + lane;
                    kk[c] = idx < ncand ? (~cur[idx]) - (K - 1) : 0;
                }
                packet_sync();
#pragma unroll
                for (int c = 0; c < kMaxList / kWave; ++c) {
                    const int idx = c * kWave + lane;
                    if (idx < ncand) s_b[idx] = kk[c];
                }
                kk0 = kk[0];
            }
            // stage the SRT records of the first 64 candidates (fast mode: kFastSlots, the table follows them): lanes
            // over candidates, one gather round trip
            if (lane < ncand && lane < (fast ? kFastSlots : kRecSlots))
                rec_to_lds(s_rec, lane, rec_from_global(pp, pr, ps, kk0), kk0);
            packet_sync();
        }
    }

    if (MVP_DEBUG_STAGE(p) == 1) ncand = 0;
    const v2f oxy = {o.x, o.y}, dxy = {d.x, d.y};
    // ---------------- exact per-ray leaf test (utils.h:744-761), lanes over rays ----------------
    float rtmin = INFINITY, rtmax = -INFINITY;
    bool ranges_ok = true;  // false when a step index does not fit the packed 16-bit range
    // lane-independent mode: list entry (k) and packed packet range of slot `lane`, and this ray's crossing list
    int ent0 = 0, rg0 = 0;
    uint32_t head = kNullLink;
    int ncross = 0;
    if (FAST && fast) {
        // Same test as below, and in addition every ray records ITS OWN crossings (slot, first step, step count) in
        // the LDS table s_tab[j * 64 + lane], j = 0,1,.. in list order, linked in order of the first step:
        //   entry = slot | next << 6 | (steps - 1) << 11 | first step << 17.
        // Records are compacted so that s_rec[slot] is the record of list slot `slot`.
        uint32_t tailj = kNullLink;
        int taillo = -1;
        bool lfail = false;
        bool wfail = false;
        for (int c = 0; c < ncand; ++c) {
            const int k = c < kWave ? __builtin_amdgcn_readlane(kk0, c) : __builtin_amdgcn_readlane(kk1, c - kWave);
            const bool inlds = c < kFastSlots;
            Rec qg;
            if (!inlds) qg = rec_from_global(pp, pr, ps, k);
            const RecP q = inlds ? recp_from_lds(s_rec, c) : recp_of(qg);
            const Y3 r0p = box_point(q, oxy, o.z), rdp = box_dir(q, dxy, d.z);  // primtransf.h:134-153
            const f3 r0 = mk3(r0p.xy.x, r0p.xy.y, r0p.z), rd = mk3(rdp.xy.x, rdp.xy.y, rdp.z);
            const f3 ird = mk3(fast_rcp(rd.x), fast_rcp(rd.y), fast_rcp(rd.z));
            const f3 t0 = mk3((-1.f - r0.x) * ird.x, (-1.f - r0.y) * ird.y, (-1.f - r0.z) * ird.z);
            const f3 t1 = mk3((1.f - r0.x) * ird.x, (1.f - r0.y) * ird.y, (1.f - r0.z) * ird.z);
            const float tn = max3f(fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z));
            const float tf = min3f(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y), fmaxf(t0.z, t1.z));
            const bool hit = active && (tn <= tf);
            if (hit) {
                rtmin = fminf(rtmin, tn);
                rtmax = fmaxf(rtmax, tf);
            }
            int lo = 0x7fffffff, hi = -1;
            const bool some = hit && lane_step_range(tn, tf, tmin, tmax, dt, lo, hi);
            if (!some) lo = 0x7fffffff, hi = -1;
            if (__ballot(some) != 0ull) {  // wave-uniform
                const int wlo = uni(wave_min(lo)), whi = uni(wave_max(hi));
                if (nh >= kFastSlots || whi > kFastMaxStep) {
                    wfail = true;
                    break;
                }
                if (lane == nh) {
                    ent0 = k;
                    rg0 = wlo | (whi << 16);
                }
                // the record moves to its list slot (nh <= c: nothing unread is overwritten; one wave, in-order LDS)
                if (inlds) {
                    if (nh != c && lane < 4) s_rec[nh * 4 + lane] = s_rec[c * 4 + lane];
                } else if (lane == 0) {
                    rec_to_lds(s_rec, nh, qg, k);
                }
                if (some) {
                    const int len = hi - lo + 1;
                    if (ncross >= kFastMaxCross || len > kFastMaxLen) {
                        lfail = true;
                    } else {
                        // sorted insert by first step; most crossings arrive in increasing order of depth within a
                        // shell, so the tail test usually avoids the walk
                        const uint32_t jn = (uint32_t)ncross;
                        uint32_t prev = kNullLink, nx = kNullLink;
                        if (lo >= taillo) {
                            prev = tailj;
                        } else {
                            uint32_t cj = head;
                            while (true) {  // ends: the tail's first step is > lo
                                const uint32_t ce = s_tab[cj * kWave + lane];
                                if ((int)(ce >> 17) > lo) {
                                    nx = cj;
                                    break;
                                }
                                prev = cj;
                                cj = (ce >> 6) & 31u;
                            }
                        }
                        s_tab[jn * kWave + lane] =
                            (uint32_t)nh | (nx << 6) | ((uint32_t)(len - 1) << 11) | ((uint32_t)lo << 17);
                        if (prev == kNullLink) {
                            head = jn;
                        } else {
                            const uint32_t pe = s_tab[prev * kWave + lane];
                            s_tab[prev * kWave + lane] = (pe & ~(31u << 6)) | (jn << 6);
                        }
                        if (nx == kNullLink) {
                            tailj = jn;
                            taillo = lo;
                        }
                        ++ncross;
                    }
                }
                ++nh;
            }
        }
        if (wfail || __ballot(lfail) != 0ull) {
            // over one of the limits: start again in slot-synchronous mode (candidates back to LDS, records re-staged)
            fast = false;
            rtmin = INFINITY, rtmax = -INFINITY;
            nh = 0;
            packet_sync();
            if (lane < ncand) s_b[lane] = kk0;
            if (lane + kWave < ncand) s_b[lane + kWave] = kk1;
            if (lane < ncand && lane < kRecSlots) rec_to_lds(s_rec, lane, rec_from_global(pp, pr, ps, kk0), kk0);
            packet_sync();
        }
    }
    for (int c = 0; c < ((FAST && fast) ? 0 : ncand); ++c) {
        const int k = uni(s_b[c]);
        const int slot = c < kRecSlots ? c : kNoSlot;
        const RecP q = (c < kRecSlots) ? recp_from_lds(s_rec, c) : recp_of(rec_from_global(pp, pr, ps, k));
        const Y3 r0p = box_point(q, oxy, o.z), rdp = box_dir(q, dxy, d.z);  // primtransf.h:134-153
        const f3 r0 = mk3(r0p.xy.x, r0p.xy.y, r0p.z), rd = mk3(rdp.xy.x, rdp.xy.y, rdp.z);
        const f3 ird = mk3(fast_rcp(rd.x), fast_rcp(rd.y), fast_rcp(rd.z));
        const f3 t0 = mk3((-1.f - r0.x) * ird.x, (-1.f - r0.y) * ird.y, (-1.f - r0.z) * ird.z);
        const f3 t1 = mk3((1.f - r0.x) * ird.x, (1.f - r0.y) * ird.y, (1.f - r0.z) * ird.z);
        const float tn = max3f(fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z));
        const float tf = min3f(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y), fmaxf(t0.z, t1.z));
        const bool hit = active && (tn <= tf);
        if (hit) {
            rtmin = fminf(rtmin, tn);
            rtmax = fmaxf(rtmax, tf);
        }
        // lattice steps of this ray that can fall inside this primitive
        int lo = 0x7fffffff, hi = -1;
        const bool some = hit && lane_step_range(tn, tf, tmin, tmax, dt, lo, hi);
        if (!some) lo = 0x7fffffff, hi = -1;
        if (__ballot(some) != 0ull) {  // wave-uniform
            const int wlo = uni(wave_min(lo)), whi = uni(wave_max(hi));
            if (whi >= 65535) ranges_ok = false;
            // (slot nh <= c of s_b is overwritten below: the read of s_b[c] above was issued earlier by this same
            //  wave -- the only one in the workgroup -- and LDS operations of a wave complete in order)
            if (nh < kMaxList) {
                if (lane == 0) {
                    s_b[nh] = k | (slot << 24);
                    s_a[nh] = min(wlo, 65535) | (min(whi, 65535) << 16);
                }
                ++nh;
            } else if (p.diag && lane == 0) {
                atomicAdd(p.diag + MVP_DIAG_LIST_OVERFLOW, 1u);
            }
        }
    }
    packet_sync();

    if (MVP_DEBUG_STAGE(p) == 2) nh = 0;
    // ---------------- grad mode: hand this packet's list to the primitive-centric backward ----------------
    if (!BWD && p.pl_count != nullptr && nh > 0) {
        uint32_t *flags = p.pl_count + (size_t)p.N * K;
        if (FAST && fast) {
            if (lane < nh) {
                const size_t pk = (size_t)n * K + ent0;
                const uint32_t idx = atomicAdd(p.pl_count + pk, 1u);
                if (idx < (uint32_t)p.pl_cap) {
                    p.pl_list[pk * (size_t)p.pl_cap + idx] = make_uint2(((uint32_t)tidx << 9) | (uint32_t)lane, (uint32_t)rg0);
                } else {
                    raise_flag(flags, kFlagListOverflow);
                    flags[3 + (size_t)n * p.tiles_x * p.tiles_y + tidx] = kPacketFwdOverflow;  // (region zeroed by the host)
                }
            }
        } else {
            for (int j = lane; j < nh; j += kWave) {
                const int k = s_b[j] & 0xffffff;
                const size_t pk = (size_t)n * K + k;
                const uint32_t idx = atomicAdd(p.pl_count + pk, 1u);
                if (idx < (uint32_t)p.pl_cap) {
                    p.pl_list[pk * (size_t)p.pl_cap + idx] = make_uint2(((uint32_t)tidx << 9) | (uint32_t)j, (uint32_t)s_a[j]);
                } else {
                    raise_flag(flags, kFlagListOverflow);
                    flags[3 + (size_t)n * p.tiles_x * p.tiles_y + tidx] = kPacketFwdOverflow;
                }
            }
            if (!ranges_ok && lane == 0) raise_flag(flags, kFlagGlobal);
        }
    }

    // ---------------- march ----------------
    rtmin = fmaxf(rtmin, tmin);  // subset_kernel.h:63-64
    rtmax = fminf(rtmax, tmax);
    const bool has = active && (rtmin < INFINITY) && nh > 0;
    const int incs = has ? (int)fminf(floorf((rtmin - tmin) * fast_rcp(dt)), 1.0e9f) : 0x7fffffff;  // subset_kernel.h:70
    const float tend = rtmax + 1e-5f;

    f3 dL3 = mk3(0.f, 0.f, 0.f);
    float dLw = 0.f;
    f3 rsat_in = mk3(-1.f, -1.f, -1.f);
    if (BWD && inimg) {
        const float4 g4 = reinterpret_cast<const float4 *>(p.grad_rayrgba)[r];  // primaccum.h:58-61
        dL3 = mk3(g4.x, g4.y, g4.z);
        dLw = g4.w;
        rsat_in = ld3(p.raysat_in + r * 3);
    }
    const bool has_sat = rsat_in.x > -1.f;  // primaccum.h:93

    if (nh > 0) {
        if (p.diag && lane == 0) {
            atomicAdd(p.diag + MVP_DIAG_PACKETS_HIT, 1u);
            atomicMax(p.diag + MVP_DIAG_MAX_LIST, (uint32_t)nh);
            atomicAdd(p.diag + MVP_DIAG_LIST_ENTRIES, (uint32_t)nh);
            atomicAdd(p.diag + MVP_DIAG_CANDIDATES, (uint32_t)ncand);
            if (!(FAST && fast)) atomicAdd(p.diag + MVP_DIAG_SLOWPATH_PACKETS, 1u);
        }
        const size_t V4 = (size_t)p.TD * p.TH * p.TW * 4;
        const float *T = p.tplate + (size_t)n * K * V4;
        float *gT = BWD ? p.grad_tplate + (size_t)n * K * V4 : nullptr;
        const int sW = 4, sH = p.TW * 4, sD = p.TH * p.TW * 4;  // float strides of the channels-last slab
        const float mx = 0.5f * (float)(p.TW - 1), my = 0.5f * (float)(p.TH - 1), mz = 0.5f * (float)(p.TD - 1);

        if (FAST && fast) {
            // ---- lane-independent sweep ----------------------------------------------------------------------
            // Every ray walks ITS OWN samples in the reference's order (lattice step ascending, list slot ascending
            // within a step: subset_kernel.h:76-97) at its own pace: no lane waits for the packet's step counter, and
            // the inside test runs only where the ray's own step range says a sample can be.  `act` = my crossings
            // that contain step s (bit j = my j-th crossing in list order), `cur` = those not yet visited at s; the
            // next crossing to open is (nj, en) in first-step order.  One loop iteration = at most one sample per lane.
            uint32_t act = 0u, cur = 0u, nj = head, en = 0u;
            int nlo = 0x7fffffff, s = 0;
            if (nj != kNullLink) {
                en = s_tab[nj * kWave + lane];
                nlo = (int)(en >> 17);
            }
            bool work = has && ncross > 0;
            v2f xxy = oxy;
            float xz = o.z;
            while (__ballot(work) != 0ull) {
                if (work) {
                    if (cur == 0u) {  // step s is done: next step with an open crossing
                        if (act == 0u && nj == kNullLink) {
                            work = false;
                        } else {
                            s = act != 0u ? s + 1 : nlo;
                            while (nlo <= s) {  // crossings that open here (first-step order)
                                act |= 1u << nj;
                                nj = (en >> 6) & 31u;
                                nlo = 0x7fffffff;
                                if (nj != kNullLink) {
                                    en = s_tab[nj * kWave + lane];
                                    nlo = (int)(en >> 17);
                                }
                            }
                            const float t = lattice_t(s, dt, tmin);
                            if (t < tend) {  // subset_kernel.h:84; t only grows from here
                                cur = act;
                                ray_point(oxy, o.z, dxy, d.z, t, xxy, xz);
                            } else {
                                work = false;
                            }
                        }
                    }
                    if (cur != 0u) {
                        const int j = __ffs((int)cur) - 1;
                        cur &= cur - 1u;
                        const uint32_t e = s_tab[j * kWave + lane];
                        const int slot = (int)(e & 63u);
                        if (s >= (int)(e >> 17) + (int)((e >> 11) & 63u)) act &= ~(1u << j);  // its last step
                        const float4 ra = s_rec[slot * 4 + 0], rb = s_rec[slot * 4 + 1], rc = s_rec[slot * 4 + 2],
                                     rd4 = s_rec[slot * 4 + 3];
                        RecP q;
                        q.r0xy = v2f{ra.x, ra.y}, q.r1xy = v2f{ra.z, ra.w}, q.r2xy = v2f{rb.x, rb.y}, q.pxy = v2f{rb.z, rb.w};
                        q.r0z = rc.x, q.r1z = rc.y, q.r2z = rc.z, q.pz = rc.w;
                        q.sxy = v2f{rd4.x, rd4.y}, q.sz = rd4.z;
                        const int k = __float_as_int(rd4.w);
                        const Y3 yp = box_point(q, xxy, xz);
                        if (s >= incs && strictly_inside(yp)) {
                            const f3 y = mk3(yp.xy.x, yp.xy.y, yp.z);
                            float4 v;
                            if constexpr (TS > 0)
                                v = sample_slab_c<FADE8, TS>(T, (uint32_t)k * (uint32_t)(TS * TS * TS * 16), y, p.fadescale,
                                                             p.fadeexp);
                            else
                                v = sample_slab<FADE8>(T + (size_t)k * V4, y, p.TD, p.TH, p.TW, p.fadescale, p.fadeexp);
                            float contrib;
                            nanw = nanw || (v.w != v.w);
                            if (composite(rgba, v, dt, contrib)) {  // saturated: nothing after this sample is evaluated
                                raysat = mk3(v.x, v.y, v.z);
                                satkey = ((uint32_t)s << 9) | (uint32_t)slot;
                                wbefore = rgba.w - contrib;
                                work = false;
                            }
                        }
                    }
                }
            }
        } else {
            // step window of the packet
            int s = uni(wave_min(incs));
            int s_last;
            {
                int mylast = -1;
                for (int j = lane; j < nh; j += kWave) mylast = max(mylast, ranges_ok ? ((s_a[j] >> 16) & 0xffff) : 0x7ffffffe);
                s_last = uni(wave_max(mylast));
                // no sample at or beyond t = tend: bound the sweep by the rays' own end as well
                const int myend = has ? (int)fminf(floorf((tend - tmin) * fast_rcp(dt)) + 1.f, 1.0e9f) : -1;
                s_last = min(s_last, uni(wave_max(myend)));
            }
            const int nchunks = (nh + kWave - 1) / kWave;
            bool sat = false;
            // list slot `lane` of chunk 0 (every packet of a head-like scene fits in it): range and entry in registers
            const int rgc0 = lane < nh ? s_a[lane] : 0;
            const int entc0 = lane < nh ? s_b[lane] : 0;

            while (s <= s_last) {
                if (__ballot(has && !sat) == 0ull) break;  // every ray saturated (subset_kernel.h:76)
                const float t = lattice_t(s, dt, tmin);
                v2f xxy;
                float xz_;
                ray_point(oxy, o.z, dxy, d.z, t, xxy, xz_);
                const f3 x = mk3(xxy.x, xxy.y, xz_);
                const bool inrange = has && s >= incs && t < tend;
                bool anyslot = false;
                int nextlo = 0x7fffffff;
                for (int ch = 0; ch < nchunks; ++ch) {
                    const int j = ch * kWave + lane;
                    bool on = false;
                    if (j < nh) {
                        const int rg = ch == 0 ? rgc0 : s_a[j];
                        const int lo = rg & 0xffff, hi = (rg >> 16) & 0xffff;
                        on = !ranges_ok || (lo <= s && s <= hi);
                        if (lo > s) nextlo = min(nextlo, lo);
                    }
                    unsigned long long m = __ballot(on);
                    anyslot = anyslot || (m != 0ull);
                    if (!BWD) {
                        // Forward: (A) every active slot's inside test with the record broadcast from LDS -> per-lane
                        // bitmask of the slots this ray is inside at this step; (B) every lane then consumes ITS OWN
                        // slots in ascending list order, all lanes sampling at once (records gathered per lane).  On
                        // head-like scenes the boxes active at one step cover mostly disjoint parts of the packet, so
                        // (B) runs ~overlap-depth rounds instead of one round per active slot.
                        unsigned long long mine = 0ull;
                        while (m) {  // list entries of chunk 0 come from registers via v_readlane (no LDS round trip)
                            const int bit = __ffsll((long long)m) - 1;
                            m &= m - 1ull;
                            const int ent = ch == 0 ? __builtin_amdgcn_readlane(entc0, bit) : uni(s_b[ch * kWave + bit]);
                            const int slot = (ent >> 24) & 0xff;
                            const RecP q = (slot != kNoSlot) ? recp_from_lds(s_rec, slot)
                                                              : recp_of(rec_from_global(pp, pr, ps, ent & 0xffffff));
                            const bool inside = inrange && !sat && strictly_inside(box_point(q, xxy, x.z));  // subset_kernel.h:84
                            if (inside) mine |= 1ull << bit;
                        }
                        if (MVP_DEBUG_STAGE(p) == 3) mine = 0ull;
                        while (__ballot(mine != 0ull) != 0ull) {
                            if (mine != 0ull) {
                                const int bit = __ffsll((long long)mine) - 1;
                                mine &= mine - 1ull;
                                const int ent = s_b[ch * kWave + bit];
                                int k = ent & 0xffffff;
                                const int slot = (ent >> 24) & 0xff;
                                // Opaque on purpose: with the TS > 0 sampler below, hipcc (ROCm 7.2) dropped this mask and fed
                                // the raw entry (slot bits included) to the 64-bit address of the record loads -> wild reads.
                                asm volatile("; k = entry & 0xffffff" : "+v"(k));
                                const RecP q = (slot != kNoSlot) ? recp_from_lds(s_rec, slot)
                                                                  : recp_of(rec_from_global(pp, pr, ps, k));
                                const Y3 yp = box_point(q, xxy, x.z);
                                const f3 y = mk3(yp.xy.x, yp.xy.y, yp.z);
                                float4 v;
                                if (WARP) {  // primsampler.h:48-63 with dowarp: fade from y0, template sampled at warp(y0)
                                    const size_t VW3 = (size_t)p.WD * p.WH * p.WW * 3;
                                    const f3 y1 = warp_lookup(p.warp + ((size_t)n * K + k) * VW3, y, p.WD, p.WH, p.WW);
                                    v = tplate_lookup_general(T + (size_t)k * V4, y1, p.TD, p.TH, p.TW);
                                    v.w *= fade_of<FADE8>(y, p.fadescale, p.fadeexp);
                                } else {
                                    if constexpr (TS > 0)
                                        v = sample_slab_c<FADE8, TS>(T, (uint32_t)k * (uint32_t)(TS * TS * TS * 16), y,
                                                                     p.fadescale, p.fadeexp);
                                    else
                                        v = sample_slab<FADE8>(T + (size_t)k * V4, y, p.TD, p.TH, p.TW, p.fadescale,
                                                               p.fadeexp);
                                }
                                float contrib;
                                nanw = nanw || (v.w != v.w);
                                if (composite(rgba, v, dt, contrib)) {
                                    raysat = mk3(v.x, v.y, v.z);
                                    sat = true;
                                    satkey = ((uint32_t)s << 9) | (uint32_t)(ch * kWave + bit);
                                    wbefore = rgba.w - contrib;
                                    mine = 0ull;  // saturated: nothing after this sample is evaluated
                                }
                            }
                        }
                        continue;
                    }
                    while (m) {
                        const int bit = __ffsll((long long)m) - 1;
                        m &= m - 1ull;
                        const int ent = uni(s_b[ch * kWave + bit]);
                        const int k = ent & 0xffffff, slot = (ent >> 24) & 0xff;
                        const Rec q = (slot != kNoSlot) ? rec_from_lds(s_rec, slot) : rec_from_global(pp, pr, ps, k);
                        const f3 xmt = x - q.pos;
                        const f3 rxmt = rot_rows(q, xmt);
                        const f3 y = rxmt * q.scale;
                        const bool inside = inrange && !sat && y.x > -1.f && y.x < 1.f && y.y > -1.f && y.y < 1.f &&
                                            y.z > -1.f && y.z < 1.f;  // primtransf.h:112-117, subset_kernel.h:84
                        if (__ballot(inside) == 0ull) continue;
                        // fallback backward: only primitives the primitive-centric kernel could not own
                        bool emit = true;
                        if (BWD && !emit_all) {
                            const uint32_t c_ = p.pl_count[(size_t)n * K + k];
                            emit = (c_ & kCountDead) != 0u || (c_ & kCountMask) > (uint32_t)p.pl_cap;
                        }

                        f3 gy = mk3(0.f, 0.f, 0.f);  // BWD: dL/dy of this lane's sample (0 when not inside)
                        if (BWD && WARP && inside) {
                            // ---- warp-field sampler, backward (primsampler.h:68-91 with dowarp; utils.h:504-643 twice) ----
                            const float fade = fade_of<FADE8>(y, p.fadescale, p.fadeexp);
                            f3 ypow;
                            if (FADE8) {
                                const f3 y2 = y * y, y4 = y2 * y2;
                                ypow = y4 * y2 * y;
                            } else {
                                const float e1 = p.fadeexp - 1.f;
                                ypow = mk3(fast_pow(fabsf(y.x), e1) * (y.x > 0.f ? 1.f : -1.f),
                                           fast_pow(fabsf(y.y), e1) * (y.y > 0.f ? 1.f : -1.f),
                                           fast_pow(fabsf(y.z), e1) * (y.z > 0.f ? 1.f : -1.f));
                            }
                            const size_t VW3 = (size_t)p.WD * p.WH * p.WW * 3;
                            const float *Wk = p.warp + ((size_t)n * K + k) * VW3;
                            const TriG tw = tri_general(y, p.WD, p.WH, p.WW);
                            f3 y1 = mk3(0.f, 0.f, 0.f);
    #pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                int vox;
                                float w;
                                if (tri_inb(tw, c, p.WD, p.WH, p.WW, vox, w)) {
                                    const float *qw = Wk + (size_t)vox * 3;
                                    y1.x += qw[0] * w, y1.y += qw[1] * w, y1.z += qw[2] * w;
                                }
                            }
                            const float *Tk = T + (size_t)k * V4;
                            const TriG tt = tri_general(y1, p.TD, p.TH, p.TW);
                            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    #pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                int vox;
                                float w;
                                if (tri_inb(tt, c, p.TD, p.TH, p.TW, vox, w)) {
                                    const float4 qv = *reinterpret_cast<const float4 *>(Tk + (size_t)vox * 4);
                                    v.x += qv.x * w, v.y += qv.y * w, v.z += qv.z * w, v.w += qv.w * w;
                                }
                            }
                            const float alpha = v.w * fade;
                            // ---- primaccum.h:81-98 ----
                            const float a = alpha * dt;
                            const bool thissat = rgba.w + a >= 1.f;
                            sat = sat || thissat;
                            const float weight = sat ? (1.f - rgba.w) : a;
                            float4 dLs;
                            dLs.x = weight * dL3.x;
                            dLs.y = weight * dL3.y;
                            dLs.z = weight * dL3.z;
                            dLs.w = sat ? 0.f
                                        : dt * ((v.x - (has_sat ? rsat_in.x : 0.f)) * dL3.x +
                                                (v.y - (has_sat ? rsat_in.y : 0.f)) * dL3.y +
                                                (v.z - (has_sat ? rsat_in.z : 0.f)) * dL3.z + (has_sat ? 0.f : dLw));
                            rgba.x += v.x * weight;
                            rgba.y += v.y * weight;
                            rgba.z += v.z * weight;
                            rgba.w += weight;
                            if (emit) {
                                const float gf = -(p.fadescale * p.fadeexp) * alpha * dLs.w;
                                gy = ypow * gf;
                                dLs.w *= fade;
                                float *gTk = gT + (size_t)k * V4;
                                f3 gi1 = mk3(0.f, 0.f, 0.f);
    #pragma unroll
                                for (int c = 0; c < 8; ++c) {
                                    int vox;
                                    float w;
                                    if (tri_inb(tt, c, p.TD, p.TH, p.TW, vox, w)) {
                                        const float4 qv = *reinterpret_cast<const float4 *>(Tk + (size_t)vox * 4);
                                        float *g = gTk + (size_t)vox * 4;
                                        atomicAdd(g + 0, w * dLs.x);
                                        atomicAdd(g + 1, w * dLs.y);
                                        atomicAdd(g + 2, w * dLs.z);
                                        atomicAdd(g + 3, w * dLs.w);
                                        tri_posgrad_acc(tt, c, qv.x * dLs.x + qv.y * dLs.y + qv.z * dLs.z + qv.w * dLs.w, gi1);
                                    }
                                }
                                const f3 g1 = mk3(mx * gi1.x, my * gi1.y, mz * gi1.z);  // dL/dy1
                                float *gWk = p.grad_warp + ((size_t)n * K + k) * VW3;
                                f3 gi0 = mk3(0.f, 0.f, 0.f);
    #pragma unroll
                                for (int c = 0; c < 8; ++c) {
                                    int vox;
                                    float w;
                                    if (tri_inb(tw, c, p.WD, p.WH, p.WW, vox, w)) {
                                        const float *qw = Wk + (size_t)vox * 3;
                                        float *g = gWk + (size_t)vox * 3;
                                        atomicAdd(g + 0, w * g1.x);
                                        atomicAdd(g + 1, w * g1.y);
                                        atomicAdd(g + 2, w * g1.z);
                                        tri_posgrad_acc(tw, c, qw[0] * g1.x + qw[1] * g1.y + qw[2] * g1.z, gi0);
                                    }
                                }
                                gy.x += 0.5f * (float)(p.WW - 1) * gi0.x;
                                gy.y += 0.5f * (float)(p.WH - 1) * gi0.y;
                                gy.z += 0.5f * (float)(p.WD - 1) * gi0.z;
                            }
                        } else if (inside) {
                            // ---- fade (primsampler.h:48-51) ----
                            float fade;
                            f3 ypow;  // |y|^(fadeexp-1) * sgn(y), backward only
                            if (FADE8) {
                                const f3 y2 = y * y, y4 = y2 * y2;
                                fade = fast_exp(-p.fadescale * (y4.x * y4.x + y4.y * y4.y + y4.z * y4.z));
                                if (BWD) ypow = y4 * y2 * y;
                            } else {
                                const f3 ay = mk3(fabsf(y.x), fabsf(y.y), fabsf(y.z));
                                fade = fast_exp(-p.fadescale * (fast_pow(ay.x, p.fadeexp) + fast_pow(ay.y, p.fadeexp) +
                                                                fast_pow(ay.z, p.fadeexp)));
                                if (BWD) {
                                    const float e1 = p.fadeexp - 1.f;
                                    ypow = mk3(fast_pow(ay.x, e1) * (y.x > 0.f ? 1.f : -1.f),
                                               fast_pow(ay.y, e1) * (y.y > 0.f ? 1.f : -1.f),
                                               fast_pow(ay.z, e1) * (y.z > 0.f ? 1.f : -1.f));
                                }
                            }
                            // ---- trilinear, align_corners=True (utils.h:414-468).  y strictly inside (-1,1) puts
                            //      i in [0, T-1]; clamping the base corner to T-2 keeps all 8 corners in bounds and
                            //      gives the same value as the reference's zero-padded form (the weight of an
                            //      out-of-bounds corner is exactly 0 there).
                            const float ix = (y.x + 1.f) * 0.5f * (float)(p.TW - 1);
                            const float iy = (y.y + 1.f) * 0.5f * (float)(p.TH - 1);
                            const float iz = (y.z + 1.f) * 0.5f * (float)(p.TD - 1);
                            const int x0 = min((int)floorf(ix), p.TW - 2), y0 = min((int)floorf(iy), p.TH - 2),
                                      z0 = min((int)floorf(iz), p.TD - 2);
                            const float wx1 = ix - (float)x0, wx0 = (float)(x0 + 1) - ix;
                            const float wy1 = iy - (float)y0, wy0 = (float)(y0 + 1) - iy;
                            const float wz1 = iz - (float)z0, wz0 = (float)(z0 + 1) - iz;
                            const size_t vbase = (size_t)k * V4 + (size_t)z0 * sD + (size_t)y0 * sH + (size_t)x0 * sW;
                            const float *Tp = T + vbase;
                            const float4 c000 = *reinterpret_cast<const float4 *>(Tp);
                            const float4 c001 = *reinterpret_cast<const float4 *>(Tp + sW);
                            const float4 c010 = *reinterpret_cast<const float4 *>(Tp + sH);
                            const float4 c011 = *reinterpret_cast<const float4 *>(Tp + sH + sW);
                            const float4 c100 = *reinterpret_cast<const float4 *>(Tp + sD);
                            const float4 c101 = *reinterpret_cast<const float4 *>(Tp + sD + sW);
                            const float4 c110 = *reinterpret_cast<const float4 *>(Tp + sD + sH);
                            const float4 c111 = *reinterpret_cast<const float4 *>(Tp + sD + sH + sW);
                            const float w000 = wx0 * wy0 * wz0, w001 = wx1 * wy0 * wz0, w010 = wx0 * wy1 * wz0,
                                        w011 = wx1 * wy1 * wz0, w100 = wx0 * wy0 * wz1, w101 = wx1 * wy0 * wz1,
                                        w110 = wx0 * wy1 * wz1, w111 = wx1 * wy1 * wz1;
                            float4 v;
                            v.x = c000.x * w000 + c001.x * w001 + c010.x * w010 + c011.x * w011 + c100.x * w100 +
                                  c101.x * w101 + c110.x * w110 + c111.x * w111;
                            v.y = c000.y * w000 + c001.y * w001 + c010.y * w010 + c011.y * w011 + c100.y * w100 +
                                  c101.y * w101 + c110.y * w110 + c111.y * w111;
                            v.z = c000.z * w000 + c001.z * w001 + c010.z * w010 + c011.z * w011 + c100.z * w100 +
                                  c101.z * w101 + c110.z * w110 + c111.z * w111;
                            v.w = c000.w * w000 + c001.w * w001 + c010.w * w010 + c011.w * w011 + c100.w * w100 +
                                  c101.w * w101 + c110.w * w110 + c111.w * w111;
                            const float alpha = v.w * fade;  // primsampler.h:63

                            {  // (only the backward instantiation reaches this body; the forward left through pass B above)
                                // ---- primaccum.h:81-98 ----
                                const float a = alpha * dt;
                                const bool thissat = rgba.w + a >= 1.f;
                                sat = sat || thissat;
                                const float weight = sat ? (1.f - rgba.w) : a;
                                float4 dLs;
                                dLs.x = weight * dL3.x;
                                dLs.y = weight * dL3.y;
                                dLs.z = weight * dL3.z;
                                dLs.w = sat ? 0.f
                                            : dt * ((v.x - (has_sat ? rsat_in.x : 0.f)) * dL3.x +
                                                    (v.y - (has_sat ? rsat_in.y : 0.f)) * dL3.y +
                                                    (v.z - (has_sat ? rsat_in.z : 0.f)) * dL3.z + (has_sat ? 0.f : dLw));
                                rgba.x += v.x * weight;
                                rgba.y += v.y * weight;
                                rgba.z += v.z * weight;
                                rgba.w += weight;
                                if (emit) {
                                // ---- primsampler.h:70-76 ----
                                const float gf = -(p.fadescale * p.fadeexp) * alpha * dLs.w;
                                gy = ypow * gf;
                                dLs.w *= fade;
                                // ---- utils.h:582-589: scatter w_c * dL to the 8 corners (32 fp32 atomics) ----
                                float *Gp = gT + vbase;
    #define MVP_SCATTER(OFF_, WGT_)                           \
        atomicAdd(Gp + (OFF_) + 0, (WGT_) * dLs.x);           \
        atomicAdd(Gp + (OFF_) + 1, (WGT_) * dLs.y);           \
        atomicAdd(Gp + (OFF_) + 2, (WGT_) * dLs.z);           \
        atomicAdd(Gp + (OFF_) + 3, (WGT_) * dLs.w);
                                MVP_SCATTER(0, w000)
                                MVP_SCATTER(sW, w001)
                                MVP_SCATTER(sH, w010)
                                MVP_SCATTER(sH + sW, w011)
                                MVP_SCATTER(sD, w100)
                                MVP_SCATTER(sD + sW, w101)
                                MVP_SCATTER(sD + sH, w110)
                                MVP_SCATTER(sD + sH + sW, w111)
    #undef MVP_SCATTER
                                // ---- utils.h:592-642: d/d(position) ----
    #define MVP_DOT4(C_) ((C_).x * dLs.x + (C_).y * dLs.y + (C_).z * dLs.z + (C_).w * dLs.w)
                                const float d000 = MVP_DOT4(c000), d001 = MVP_DOT4(c001), d010 = MVP_DOT4(c010),
                                            d011 = MVP_DOT4(c011), d100 = MVP_DOT4(c100), d101 = MVP_DOT4(c101),
                                            d110 = MVP_DOT4(c110), d111 = MVP_DOT4(c111);
    #undef MVP_DOT4
                                const float gix = wy0 * wz0 * (d001 - d000) + wy1 * wz0 * (d011 - d010) +
                                                  wy0 * wz1 * (d101 - d100) + wy1 * wz1 * (d111 - d110);
                                const float giy = wx0 * wz0 * (d010 - d000) + wx1 * wz0 * (d011 - d001) +
                                                  wx0 * wz1 * (d110 - d100) + wx1 * wz1 * (d111 - d101);
                                const float giz = wx0 * wy0 * (d100 - d000) + wx1 * wy0 * (d101 - d001) +
                                                  wx0 * wy1 * (d110 - d010) + wx1 * wy1 * (d111 - d011);
                                gy.x += mx * gix;
                                gy.y += my * giy;
                                gy.z += mz * giz;
                                }  // emit
                            }
                        }
                        if (BWD && emit) {
                            // ---- primtransf.h:155-179.  grad_scale_j = sum rxmt_j*gy_j, grad_R[i][j] = s_j * sum xmt_i*gy_j,
                            //      grad_pos_i = -sum_j R[i][j]*s_j * sum gy_j: 12 wave sums, then 15 lanes flush. ----
                            // lanes without a sample contribute exact zeros (their x may be inf/NaN: rays outside the image)
                            const f3 xm = inside ? xmt : mk3(0.f, 0.f, 0.f);
                            const float a0 = uni(wave_sum(gy.x)), a1 = uni(wave_sum(gy.y)), a2 = uni(wave_sum(gy.z));
                            const float c00 = uni(wave_sum(xm.x * gy.x)), c01 = uni(wave_sum(xm.x * gy.y)),
                                        c02 = uni(wave_sum(xm.x * gy.z));
                            const float c10 = uni(wave_sum(xm.y * gy.x)), c11 = uni(wave_sum(xm.y * gy.y)),
                                        c12 = uni(wave_sum(xm.y * gy.z));
                            const float c20 = uni(wave_sum(xm.z * gy.x)), c21 = uni(wave_sum(xm.z * gy.y)),
                                        c22 = uni(wave_sum(xm.z * gy.z));
                            float val = 0.f;
                            float *dst = nullptr;
                            const f3 sa = mk3(q.scale.x * a0, q.scale.y * a1, q.scale.z * a2);
                            switch (lane) {
                                case 0: val = q.scale.x * c00; break;
                                case 1: val = q.scale.y * c01; break;
                                case 2: val = q.scale.z * c02; break;
                                case 3: val = q.scale.x * c10; break;
                                case 4: val = q.scale.y * c11; break;
                                case 5: val = q.scale.z * c12; break;
                                case 6: val = q.scale.x * c20; break;
                                case 7: val = q.scale.y * c21; break;
                                case 8: val = q.scale.z * c22; break;
                                case 9: val = q.r0.x * c00 + q.r1.x * c10 + q.r2.x * c20; break;   // sum rxmt_x * gy_x
                                case 10: val = q.r0.y * c01 + q.r1.y * c11 + q.r2.y * c21; break;
                                case 11: val = q.r0.z * c02 + q.r1.z * c12 + q.r2.z * c22; break;
                                case 12: val = -dot3(q.r0, sa); break;
                                case 13: val = -dot3(q.r1, sa); break;
                                case 14: val = -dot3(q.r2, sa); break;
                                default: break;
                            }
                            if (lane < 9)
                                dst = p.grad_primrot + ((size_t)n * K + k) * 9 + lane;
                            else if (lane < 12)
                                dst = p.grad_primscale + ((size_t)n * K + k) * 3 + (lane - 9);
                            else if (lane < 15)
                                dst = p.grad_primpos + ((size_t)n * K + k) * 3 + (lane - 12);
                            if (dst) atomicAdd(dst, val);
                        }
                    }
                }
                if (anyslot) {
                    ++s;
                } else {  // nothing listed covers this step: jump to the next range start
                    const int nx = uni(wave_min(nextlo));
                    if (nx == 0x7fffffff) break;
                    s = nx;
                }
            }
        }
    }

    if (!BWD && p.pl_count != nullptr) {
        // max |raysat| over the packet -> tail word [2] (the backward's fixed-point bound).  |-1| = 1 when unsaturated:
        // the host pre-sets the word to 1.0f, and only a packet that can raise it touches it.  (One same-address
        // atomic per packet -- 327 680 of them at C2 -- serialised in L2 and cost 2.5 ms of a 9.4 ms kernel.)
        float m = inimg ? fmaxf(fabsf(raysat.x), fmaxf(fabsf(raysat.y), fabsf(raysat.z))) : 0.f;
        if (!(m == m)) m = INFINITY;
        m = uni(wave_max(m));
        if (__ballot(nanw) != 0ull && lane == 0) raise_flag(p.pl_count + (size_t)p.N * K, kFlagGlobal);
        if (m > 1.0f) {
            uint32_t *word = p.pl_count + (size_t)p.N * K + 2;
            const float cur = __uint_as_float(__atomic_load_n(word, __ATOMIC_RELAXED));  // stale is fine: monotone
            if (m > cur && lane == 0) atomicMax(word, __float_as_uint(m));
        }
    }
    if (!BWD && inimg) {
        reinterpret_cast<float4 *>(p.rayrgba)[r] = rgba;  // primaccum.h:51-56
        if (p.raysat) {
            float *sp = p.raysat + r * 3;
            MVP_STREAM_STOREF(sp, raysat.x), MVP_STREAM_STOREF(sp + 1, raysat.y), MVP_STREAM_STOREF(sp + 2, raysat.z);
        }
        if (p.rayaux)
            MVP_STREAM_STORE(reinterpret_cast<float4 *>(p.rayaux) + r,
                             make_float4(__uint_as_float(satkey), wbefore, __uint_as_float((uint32_t)incs), tend));
    }
}

// TS > 0 (forward, no warp field): TS^3 slabs with compile-time strides, see sample_slab_c
template <bool BWD, bool FADE8, bool WARP, int TS = 0>
#ifdef MVP_FWD_OCC  // timing variants only (profiles/r04_forward_experiments.json: six waves per SIMD)
__global__ __launch_bounds__(kWave, MVP_FWD_OCC) void march_kernel(const MarchParams p) {
#else
__global__ __launch_bounds__(BWD ? kWave : kWave * MVP_FWD_QUAD) void march_kernel(const MarchParams p) {
#endif
    // One LDS block per wave: [SRT records: 64 x 64 B][region].  The region is the two 512-entry frontier / list arrays
    // (s_a, s_b); in the plain forward it is large enough to be re-used, after the traversal, as the per-ray crossing
    // table of the lane-independent sweep (kFastCross rows x 64 lanes x 4 B).
    // slot-synchronous layout: 64 records + s_a + s_b; lane-independent layout: kFastSlots records + kFastCross rows
    constexpr int kSlowWords = kRecSlots * 16 + 2 * kMaxList;
    constexpr int kFastWords = kFastSlots * 16 + kFastCross * kWave;
    constexpr int kWords = (!BWD && !WARP && kFastWords > kSlowWords) ? kFastWords : kSlowWords;
    constexpr int kQuad = BWD ? 1 : MVP_FWD_QUAD;  // packets (waves) per workgroup
    __shared__ __attribute__((aligned(16))) uint32_t smem_all[kWords * kQuad];
    const int qwave = kQuad > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    uint32_t *smem = smem_all + qwave * kWords;
    float4 *s_rec = reinterpret_cast<float4 *>(smem);
    int *s_a = reinterpret_cast<int *>(smem + kRecSlots * 16);
    int *s_b = s_a + kMaxList;
    uint32_t *s_tab = smem + kFastSlots * 16;
    if (BWD) {
        bool emit_all = p.fallback_all != 0;
        if (!emit_all) {  // nothing to do unless the forward raised a flag
            const uint32_t flags = p.pl_count[(size_t)p.N * p.K] & ~kFlagBwdPrecise;  // (those are not this kernel's)
            if (flags == 0u) return;
            emit_all = (flags & kFlagGlobal) != 0u;
        }
        for (int b = blockIdx.x; b < p.total_packets; b += gridDim.x) {
            march_packet<BWD, FADE8, WARP, TS>(p, b, s_a, s_b, s_rec, s_tab, emit_all);
            __syncthreads();
        }
    } else if (kQuad == 1) {
        march_packet<BWD, FADE8, WARP, TS>(p, blockIdx.x, s_a, s_b, s_rec, s_tab, true);
    } else {
        // block b runs on XCD b % 8: keep those three bits, the workgroup's waves take kQuad consecutive slots of that XCD's
        // share (consecutive slots of a strip are neighbouring packets, packet_of_block)
        const int ii = kQuad * ((int)blockIdx.x >> 3) + qwave;
        if (ii < (p.total_packets >> 3))
            march_packet<BWD, FADE8, WARP, TS>(p, ((int)blockIdx.x & 7) + 8 * ii, s_a, s_b, s_rec, s_tab, true);
    }
}


// =================================================================================================
// Primitive-centric backward.  One workgroup (4 waves) per (image n, primitive k).
//   LDS: [V] float4 template slab | [4][Vp] int32 fixed-point gradient |
//        ray queue (kQueueCap x 8 B) | small reduce area.   Vp = padded voxel count (z stride TH*TW + kGradPadZ, see below).
//   Work proceeds in rounds of 8 list entries (ray packets):
//     phase 1 (lanes = the packet's rays): exact ray/box interval -> rays that really cross the box are
//             COMPACTED into the LDS queue (ballot + popcount prefix inside the wave, one LDS integer atomic per
//             wave for the queue tail).  On head-like scenes only ~40 % of a packet's rays cross a given box.
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
// sample, and the LDS pipe was busy 82 % of the kernel):  acc += rn(value * s),  s = 0.999 * 2^31 / (n * B)  where
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
//     the values are far below the bound.  A primitive whose sum_r n_r^3 passes kNoiseBudget * V (noise 3e-6 of the
//     bound, rms) is therefore handed to the two-pass instantiation BEFORE the round that would pass it is marched.
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
#ifndef MVP_EXP
// Timing experiments of the primitive-centric backward (tools/exp_variants.sh builds them, never the product library):
//   1 = conflict-free scatter addresses, 2 = no scatter atomics, 3 = no march at all (front-end chain: staging, phase 1,
//   queue, phase-2 ray loads, output) -- all compute WRONG gradients: time only,
//   4 = count same-address / same-bank lanes per 32-lane group into diag (tools/exp4_stats.py),
//   6 = fp32 LDS atomics in the scatter instead of conversion + integer atomics (WRONG gradients: time only).
#define MVP_EXP 0
#endif
// Waves per workgroup (one workgroup = one primitive) is a template parameter of the kernel, PW in {2, 3}.  The kernel is
// latency-bound per workgroup and holds ~164 VGPRs (12 waves per CU): 3 waves x 4 workgroups per CU keeps one more
// primitive in flight than round 1's 4 x 3, 2 waves x 5 workgroups two more.  Which is faster depends on how much work a
// primitive has: K = 16384 at 512^2 (few packets per primitive) prefers 2 waves (C3 backward 0.87 -> 0.74 ms), K = 8192 at
// 1024^2 prefers 3 (C4 1.90 vs 2.01 ms), C2 is indifferent; the host picks by packets per primitive.
constexpr float kFixRange = 0.999f * 2147483648.f;  // |sum of a round's contributions * scale| stays below 2^31
constexpr float kTwoPassRatio = 256.f;  // marched rays' gradient magnitudes further below the bound than this: two passes
constexpr float kNoiseBudget = 6.2e7f;  // sum over rounds of (samples of the round)^3 per voxel: beyond it, two passes (header)
constexpr int kRoundBudgetLog2 = 14;   // a round takes list entries while 64 lanes x their step ranges stay below 2^14
#ifndef MVP_GRADPAD
#define MVP_GRADPAD 5
#endif
constexpr int kGradPadZ = MVP_GRADPAD;  // see the note on the gradient arrays above
#ifndef MVP_ENTRIES_PER_WAVE
#define MVP_ENTRIES_PER_WAVE 5
#endif
constexpr int kEntriesPerWave = MVP_ENTRIES_PER_WAVE;  // list entries (packets) each wave examines per round
__host__ __device__ constexpr int prim_entries_per_round(int pw) { return pw * kEntriesPerWave; }  // typical lists: ONE round
__host__ __device__ constexpr int prim_queue_cap(int pw) { return prim_entries_per_round(pw) * 64; }  // rays per round
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

__device__ __forceinline__ uint32_t abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }

// float -> int, round to nearest (ties up): v_cvt_rpi_i32_f32.  (int)x truncates toward zero, a systematic shrink of
// every contribution by half a unit on average.
__device__ __forceinline__ int fix_rn(float v) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// TS > 0: the slab is TS^3 (compile-time strides: the 32 atomics and 8 reads of a sample share ONE address register and
// use immediate offsets); TS == 0: any slab size, strides in registers.
// WARP: the warp-field sampler (algo 1, primsampler.h:53-58,82-88): a second LDS slab (the warp grid) and a second set of
// fixed-point accumulators (grad_warp); the template is sampled at warp(y) with zero padding (general strides only).
#ifndef MVP_BWD_OCC
#define MVP_BWD_OCC 3  // waves per SIMD the register allocation aims at (timing experiments: 4 = at most 128 VGPRs)
#endif
// RESID: the two-pass instantiation (header, DYNAMIC RANGE): owns the primitives the plain one marked, nothing else.
template <bool FADE8, int TS, int PW, bool WARP, bool RESID>
__device__ __forceinline__ void bwd_prim_body(const MarchParams &p, const int block, float4 *smem4) {
    static_assert(!WARP || TS == 0, "the warp-field variant uses run-time slab dimensions");
    constexpr int kPrimWaves = PW, kPrimBlock = PW * 64;
    constexpr int kEntriesPerRound = prim_entries_per_round(PW), kQueueCap = prim_queue_cap(PW);
    const int TD = TS ? TS : p.TD, TH = TS ? TS : p.TH, TW = TS ? TS : p.TW;
    const int V = TD * TH * TW;
    const int gH = TW, gD = TH * TW + kGradPadZ;  // gradient-array strides (words); x stride 1
    const int Vp = TD * gD;
    float4 *s_T = smem4;
    int *s_acc = reinterpret_cast<int *>(smem4 + V);  // [4][Vp], channel-planar fixed-point sums
    uint2 *s_q = reinterpret_cast<uint2 *>(s_acc + 4 * Vp);  // (Vp is even: 8-byte aligned)
    float *s_red = reinterpret_cast<float *>(s_q + kQueueCap);  // 64 floats
    uint32_t *s_qn = reinterpret_cast<uint32_t *>(s_red + 64);
    uint32_t *s_bucket = s_qn + 4;  // kLenBuckets words
    uint16_t *s_perm = reinterpret_cast<uint16_t *>(s_bucket + kLenBuckets);  // pl_cap entries: list index by rank
    uint32_t *s_gext = reinterpret_cast<uint32_t *>(s_red + 62);  // per round: bits(max), bits(min) of the queued rays' max |g|
    // WARP: [warp grid as float4 (x,y,z,-)][3][VWp] fixed-point sums -- behind everything else (16-byte aligned: the
    // host sizes the part above as a multiple of 16 bytes)
    const int WD = WARP ? p.WD : 2, WH = WARP ? p.WH : 2, WW = WARP ? p.WW : 2;
    const int VW = WD * WH * WW, gHw = WW, gDw = WH * WW + kGradPadZ, VWp = WD * gDw;
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
    const uint2 *list = p.pl_list + pk * (size_t)p.pl_cap;
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
    const uint32_t epr = min((uint32_t)kEntriesPerRound, max(1u, (1u << kRoundBudgetLog2) / (64u * maxlen)));
    const bool multi = cnt > epr;  // more than one round: walk the entries in ascending key order (see the header)
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
    const float nfs_log2e = -p.fadescale * 1.44269504088896341f;  // exp(-fadescale * e) = exp2(nfs_log2e * e): one multiply
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    float c00 = 0.f, c01 = 0.f, c02 = 0.f, c10 = 0.f, c11 = 0.f, c12 = 0.f, c20 = 0.f, c21 = 0.f, c22 = 0.f;

#if MVP_EXP == 4
    uint32_t ex_groups = 0u, ex_addr = 0u, ex_lanes = 0u, ex_bank[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};  // wave-uniform
#endif
    if (tid == 0) s_qn[2] = 0u;
    if (multi) {
        // rank of every entry among the list's keys ((packet << 9) | slot: one entry per packet, all different); the key
        // stream is wave-uniform -> scalar loads, four entries (32 bytes; pl_cap is a multiple of 4) at a time
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // (a native vector: HIP's uint4 class has no
        typedef const __attribute__((address_space(4))) u32x4 *cu4;   //  constructor from another address space)
        const cu4 l4 = reinterpret_cast<cu4>(reinterpret_cast<uintptr_t>(list));
        for (uint32_t e = tid; e < cnt; e += kPrimBlock) {
            const uint32_t key = list[e].x;
            uint32_t rank = 0u;
            for (uint32_t j = 0; j < cnt; j += 4u) {
                const u32x4 a = l4[j >> 1], b = l4[(j >> 1) + 1];
                rank += (a.x < key ? 1u : 0u) + ((j + 1u < cnt && a.z < key) ? 1u : 0u) +
                        ((j + 2u < cnt && b.x < key) ? 1u : 0u) + ((j + 3u < cnt && b.z < key) ? 1u : 0u);
            }
            s_perm[rank] = (uint16_t)e;
        }
    }
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
            g.x = (float)s_acc[gv] * i_rgb;
            g.y = (float)s_acc[Vp + gv] * i_rgb;
            g.z = (float)s_acc[2 * Vp + gv] * i_rgb;
            g.w = (float)s_acc[3 * Vp + gv] * i_a;
            if (drained) {
                const float4 o_ = gd[v];
                g.x += o_.x, g.y += o_.y, g.z += o_.z, g.w += o_.w;
            }
            gd[v] = g;
#pragma unroll
            for (int c = 0; c < 4; ++c) s_acc[c * Vp + gv] = 0;
        }
        if constexpr (WARP) {
            float *gWd = p.grad_warp + pkd * (size_t)VW * 3;
            const int sDw = WH * WW;
            for (int v = td; v < VW; v += kPrimBlock) {
                const int z = v / sDw, rem = v - z * sDw;
                const int gv = z * gDw + rem;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    float g = (float)s_wacc[j * VWp + gv] * i_w;
                    if (drained) g += gWd[v * 3 + j];
                    gWd[v * 3 + j] = g;
                    s_wacc[j * VWp + gv] = 0;
                }
            }
        }
        drained = true;
        __syncthreads();
    };
    bool pass_b = false;  // this iteration re-marches the round it has just marched, for the residuals (workgroup-uniform)
    float noise_cube = 0.f;  // sum over the rounds so far of (samples of the round)^3 (workgroup-uniform; header)
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
        // Each wave owns up to kEntriesPerWave entries of the round; a live ray takes a ticket in the bucket of its step count
        // (LDS integer atomic), buckets are prefix-summed, and the ray is written at its sorted position, so the
        // 64 rays a wave marches together have (nearly) the same number of steps.
        const uint32_t eend = min(cnt, ebase + epr);
        uint2 item[kEntriesPerWave];  // {ray index inside the image | list slot << 23, first step | steps << 16}
        uint32_t ticket[kEntriesPerWave];
        bool live2[kEntriesPerWave];
        bool toolong = false;
        uint32_t mylen = 0u;
        uint32_t gq_hi = 0u;  // bits of the largest max |grad_rayrgba| over the packets this wave queued rays of (wave-uniform)
#pragma unroll
        for (int u = 0; u < kEntriesPerWave; ++u) {
            const uint32_t e = ebase + wave + u * (kPrimBlock / kWave);
            live2[u] = false;
            ticket[u] = 0u;
            item[u] = make_uint2(0u, 0u);
            if (e < eend) {
                const uint32_t le = multi ? (uint32_t)uni((int)s_perm[e]) : e;
                const uint32_t *lw = reinterpret_cast<const uint32_t *>(list + le);  // wave-uniform: scalar loads
                const uint2 ent = make_uint2(cload(lw), cload(lw + 1));
                const int tidx = (int)(ent.x >> 9);
                const uint32_t slot = ent.x & 511u;
                const int elo = (int)(ent.y & 0xffffu), ehi = (int)(ent.y >> 16);
                const int ty = tidx / p.tiles_x, tx = tidx - ty * p.tiles_x;
                const int px = tx * kTile + (lane & 7), py = ty * kTile + (lane >> 3);
                const bool inimg = px < p.W && py < p.H;
                const uint32_t r = inimg ? (uint32_t)py * (uint32_t)p.W + (uint32_t)px : 0u;  // index inside image n
                int slo = 1, shi = 0;
                if (inimg) {
                    // byte offsets computed in 32 bits: "SGPR base + zero-extended VGPR offset" addressing
                    const f3 o = ld3(at_bytes<float>(raypos_n, r * 12u)), d = ld3(at_bytes<float>(raydir_n, r * 12u));
                    const float2 tt = *at_bytes<float2>(tminmax_n, r * 8u);
                    const int incs = (int)*at_bytes<uint32_t>(aux_n, r * 16u + 8u);
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
                        shi = min(h0, ehi);
                    }
                }
                if (__ballot(slo <= shi) != 0ull) gq_hi = max(gq_hi, (cload(pmax_n + tidx) & kPacketMaxMask) << 2);
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
            const float wl = wave_sum((float)mylen);
            if (lane == 0 && wl > 0.f) {
                atomicAdd(s_qn + 1, (uint32_t)wl);
                atomicMax(s_gext, gq_hi);
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
            res_mul = kFixRange / (float)round_samples;
            s_rgb = uni(res_mul / Brgb), s_a = uni(res_mul / Ba);
            if constexpr (WARP) s_w = uni(res_mul / fmaxf(Bw, 1.0e-30f));
        }
        if constexpr (!RESID) {
            // (rare: header, ACCUMULATED ROUNDING) this round would take the sums' rounding noise past the budget: the
            // two-pass instantiation owns the primitive and overwrites every output (workgroup-uniform exit)
            if (!pass_b) noise_cube += (float)round_samples * (float)round_samples * (float)round_samples;
            if (noise_cube > kNoiseBudget * (float)V && s_qn[2] == 0u && !bad_bound) {
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
#if MVP_EXP == 3
            // timing experiment: everything but the march itself (front-end chain only; the loaded values stay "used")
            const int nsteps = (dL3.x + dLw + rsat.y + wbefore + tend + d.x + o.x + tmin == 12345.678f && satkey == 77u) ? 1 : 0;
#else
            const int nsteps = uni(wave_max(len));
#endif
            float ra0 = 0.f, ra1 = 0.f, ra2 = 0.f, rb0 = 0.f, rb1 = 0.f, rb2 = 0.f;
#if MVP_EXP == 2
            int exp_sink = 0;
#endif
#ifndef MVP_NO_STEP_ROTATION
            // Queue neighbours are neighbouring pixels: at the same step index they sit in the same slab cell and their
            // 64 atomics hit the same addresses (serialised by the LDS).  The samples of a ray are independent here
            // (the forward recorded where the ray saturated), so each lane walks its steps from a different starting
            // offset, wrapping around: neighbours are then at different depths at any one time.
#ifndef MVP_ROT_MUL
#define MVP_ROT_MUL 5u
#define MVP_ROT_MASK 7u
#endif
            int rot = have ? (int)(((uint32_t)ql * MVP_ROT_MUL) & MVP_ROT_MASK) : 0;
            while (rot >= len && len > 0) rot -= len;
#else
            const int rot = 0;
#endif
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
#if MVP_EXP == 4
                {   // per 32-lane group: max number of lanes on one address / on one bank (= LDS cycles of one atomic),
                    // for the current layout and for candidate (y stride, z stride mod 32) pairs
                    int code = -1;
                    if (inside) {
                        const float jx = (y.x + 1.f) * 0.5f * (float)(TW - 1), jy = (y.y + 1.f) * 0.5f * (float)(TH - 1),
                                    jz = (y.z + 1.f) * 0.5f * (float)(TD - 1);
                        code = min((int)floorf(jx), TW - 2) | (min((int)floorf(jy), TH - 2) << 4) |
                               (min((int)floorf(jz), TD - 2) << 8);
                    }
                    constexpr int NH = 8;
                    const int hy[NH] = {8, 8, 8, 8, 9, 9, 10, 12}, hz[NH] = {4, 5, 12, 20, 17, 5, 20, 3};
                    int na = 0, nb[NH];
                    for (int h = 0; h < NH; ++h) nb[h] = 0;
                    const int cx = code & 15, cy = (code >> 4) & 15, cz = (code >> 8) & 15;
                    for (int j = 0; j < 32; ++j) {
                        const int o0 = __builtin_amdgcn_readlane(code, j), o1 = __builtin_amdgcn_readlane(code, j + 32);
                        const int o = lane < 32 ? o0 : o1;
                        if (code >= 0 && o >= 0) {
                            na += (o == code) ? 1 : 0;
                            const int ox = o & 15, oy = (o >> 4) & 15, oz = (o >> 8) & 15;
#pragma unroll
                            for (int h = 0; h < NH; ++h)
                                nb[h] += (((ox - cx) + hy[h] * (oy - cy) + hz[h] * (oz - cz)) & 31) == 0 ? 1 : 0;
                        }
                    }
                    const int a0 = uni(wave_max(lane < 32 ? na : 0)), a1 = uni(wave_max(lane < 32 ? 0 : na));
                    ex_groups += (a0 > 0) + (a1 > 0);
                    ex_addr += (uint32_t)(a0 + a1);
                    ex_lanes += (uint32_t)__popcll(__ballot(inside));
#pragma unroll
                    for (int h = 0; h < NH; ++h)
                        ex_bank[h] += (uint32_t)(uni(wave_max(lane < 32 ? nb[h] : 0)) + uni(wave_max(lane < 32 ? 0 : nb[h])));
                }
#endif
                if constexpr (WARP) {
                    // ---- warp-field sampler (primsampler.h:68-91 with dowarp; utils.h:504-643 twice) ----
                    if (inside) {
                        const float fade = fade_of<FADE8>(y, p.fadescale, p.fadeexp);
                        f3 ypow;
                        if (FADE8) {
                            const f3 y2 = y * y, y4 = y2 * y2;
                            ypow = y4 * y2 * y;
                        } else {
                            const float e1 = p.fadeexp - 1.f;
                            ypow = mk3(fast_pow(fabsf(y.x), e1) * (y.x > 0.f ? 1.f : -1.f),
                                       fast_pow(fabsf(y.y), e1) * (y.y > 0.f ? 1.f : -1.f),
                                       fast_pow(fabsf(y.z), e1) * (y.z > 0.f ? 1.f : -1.f));
                        }
                        const TriG tw = tri_general(y, WD, WH, WW);  // y strictly inside: all 8 corners in bounds
                        f3 y1 = mk3(0.f, 0.f, 0.f);
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            int vox;
                            float w;
                            if (tri_inb(tw, c, WD, WH, WW, vox, w)) {
                                const float4 qw = s_W[vox];
                                y1.x += qw.x * w, y1.y += qw.y * w, y1.z += qw.z * w;
                            }
                        }
                        const TriG tt = tri_general(y1, TD, TH, TW);  // may leave the slab: zero padding
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            int vox;
                            float w;
                            if (tri_inb(tt, c, TD, TH, TW, vox, w)) {
                                const float4 qv = s_T[vox];
                                v.x += qv.x * w, v.y += qv.y * w, v.z += qv.z * w, v.w += qv.w * w;
                            }
                        }
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
#define MVP_FIXW(ACC_, IDX_, VAL_)                                                                  \
    {                                                                                               \
        const float x_ = (VAL_);                                                                    \
        const int t_ = fix_rn(x_);                                                                  \
        atomicAdd((ACC_) + (IDX_), (!RESID || !pass_b) ? t_ : fix_rn((x_ - (float)t_) * res_mul));  \
    }
                        f3 gi1 = mk3(0.f, 0.f, 0.f);
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            int vox;
                            float w;
                            if (tri_inb(tt, c, TD, TH, TW, vox, w)) {
                                const float4 qv = s_T[vox];
                                const int gv = (tt.z0 + (c >> 2)) * gD + (tt.y0 + ((c >> 1) & 1)) * gH + tt.x0 + (c & 1);
                                MVP_FIXW(s_acc, gv, w * dLs.x * s_rgb)
                                MVP_FIXW(s_acc, Vp + gv, w * dLs.y * s_rgb)
                                MVP_FIXW(s_acc, 2 * Vp + gv, w * dLs.z * s_rgb)
                                MVP_FIXW(s_acc, 3 * Vp + gv, w * dLs.w * s_a)
                                tri_posgrad_acc(tt, c, qv.x * dLs.x + qv.y * dLs.y + qv.z * dLs.z + qv.w * dLs.w, gi1);
                            }
                        }
                        const f3 g1 = mk3(mx * gi1.x, my * gi1.y, mz * gi1.z);  // dL/dy1
                        f3 gi0 = mk3(0.f, 0.f, 0.f);
#pragma unroll
                        for (int c = 0; c < 8; ++c) {
                            int vox;
                            float w;
                            if (tri_inb(tw, c, WD, WH, WW, vox, w)) {
                                const float4 qw = s_W[vox];
                                const int gv = (tw.z0 + (c >> 2)) * gDw + (tw.y0 + ((c >> 1) & 1)) * gHw + tw.x0 + (c & 1);
                                MVP_FIXW(s_wacc, gv, w * g1.x * s_w)
                                MVP_FIXW(s_wacc, VWp + gv, w * g1.y * s_w)
                                MVP_FIXW(s_wacc, 2 * VWp + gv, w * g1.z * s_w)
                                tri_posgrad_acc(tw, c, qw.x * g1.x + qw.y * g1.y + qw.z * g1.z, gi0);
                            }
                        }
#undef MVP_FIXW
                        gy.x += 0.5f * (float)(WW - 1) * gi0.x;
                        gy.y += 0.5f * (float)(WH - 1) * gi0.y;
                        gy.z += 0.5f * (float)(WD - 1) * gi0.z;
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
#if MVP_EXP == 1
                        const int gb = (lane & 31) + 32 * (st % 14);
#else
                        const int gb = (int)fmaf(fz0, (float)(gD - sD), vbf);  // z0 * gD + y0 * gH + x0 (gH = sH)
#endif
                        int *Ap = s_acc + gb;
#if MVP_EXP == 2
#define MVP_FIX1(OFF_, VAL_) exp_sink ^= fix_rn(VAL_) + (int)(OFF_);
#define MVP_FIX1B(OFF_, VAL_) MVP_FIX1(OFF_, VAL_)
#else
#if MVP_EXP == 6
// timing build: fp32 LDS atomics (ds_add_f32), no conversion -- the flush still reads integers: WRONG gradients, time only
#define MVP_FIX1(OFF_, VAL_) __hip_atomic_fetch_add(reinterpret_cast<float *>(Ap) + (OFF_), (VAL_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#elif MVP_EXP == 7
// timing build: integer atomics on the raw float bits (no conversion, no float atomic): WRONG gradients, time only
#define MVP_FIX1(OFF_, VAL_) atomicAdd(Ap + (OFF_), __float_as_int(VAL_));
#else
#define MVP_FIX1(OFF_, VAL_) atomicAdd(Ap + (OFF_), fix_rn(VAL_));
#endif
// pass B of a two-pass round: what pass A rounded away, x - rn(x) (exact in fp32), at res_mul units per unit
#define MVP_FIX1B(OFF_, VAL_)                                                   \
    {                                                                           \
        const float x_ = (VAL_);                                                \
        atomicAdd(Ap + (OFF_), fix_rn((x_ - (float)fix_rn(x_)) * res_mul));     \
    }
#endif
#define MVP_LSCATTER(FIX_, OFF_, MUL_, WP_)                         \
    {                                                               \
        const v2f a_ = MUL_(qxy, WP_), b_ = MUL_(qzw, WP_);         \
        FIX_((OFF_), a_.x)                                          \
        FIX_((OFF_) + Vp, a_.y)                                     \
        FIX_((OFF_) + 2 * Vp, b_.x)                                 \
        FIX_((OFF_) + 3 * Vp, b_.y)                                 \
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
                    }
#if MVP_EXP == 2
                    if (exp_sink == 0x12345678) s_acc[gD] = exp_sink;
#endif
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
                flush_sums(1.0f / s_rgb, 1.0f / s_a, 1.0f / s_w);
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
            const float im = 1.0f / cur_mul;
            flush_sums(im / s_rgb, im / s_a, im / s_w);
        }
        pass_b = false;
        ebase += epr;
    }
#if MVP_EXP == 4
    if (p.diag && lane == 0) {
        atomicAdd(p.diag + 0, ex_groups);
        atomicAdd(p.diag + 1, ex_addr);
        atomicAdd(p.diag + 2, ex_lanes);
        for (int h = 0; h < 8; ++h) atomicAdd(p.diag + 8 + h, ex_bank[h]);
    }
#endif
    // ---- pose gradients: 12 sums per lane -> wave -> workgroup (primtransf.h:155-179) ----
    {
        const float sums[12] = {wave_sum(a0),  wave_sum(a1),  wave_sum(a2),  wave_sum(c00), wave_sum(c01), wave_sum(c02),
                                wave_sum(c10), wave_sum(c11), wave_sum(c12), wave_sum(c20), wave_sum(c21), wave_sum(c22)};
        if (lane == 0) {
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
        const float i_rgb = 1.0f / (s_rgb * cur_mul), i_a = 1.0f / (s_a * cur_mul);
        for (int v = tl; v < V; v += kPrimBlock) {
            const int z = v / sD, rem = v - z * sD;
            const int gv = z * gD + rem;  // (y * TW + x) is the same in both layouts
            float4 g;
            g.x = (float)s_acc[gv] * i_rgb;
            g.y = (float)s_acc[Vp + gv] * i_rgb;
            g.z = (float)s_acc[2 * Vp + gv] * i_rgb;
            g.w = (float)s_acc[3 * Vp + gv] * i_a;
            if (drained) {  // (workgroup-uniform) earlier flushes sit in grad_template already; same owner thread
                const float4 o_ = gT4l[v];
                g.x = o_.x + g.x, g.y = o_.y + g.y, g.z = o_.z + g.z, g.w = o_.w + g.w;
            }
            MVP_STREAM_STORE(gT4l + v, g);
        }
    }
    if constexpr (WARP) {  // grad_warp, written exactly once
        const float i_w = 1.0f / (s_w * cur_mul);
        float *gW = p.grad_warp + pkl * (size_t)VW * 3;
        const int sDw = WH * WW;
        for (int v = tl; v < VW; v += kPrimBlock) {
            const int z = v / sDw, rem = v - z * sDw;
            const int gv = z * gDw + rem;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float g = (float)s_wacc[j * VWp + gv] * i_w;
                if (drained) g += gW[v * 3 + j];
                gW[v * 3 + j] = g;
            }
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
__global__ __launch_bounds__(PW * 64, WARP ? 2 : MVP_BWD_OCC) void bwd_prim_kernel(const MarchParams p) {
    extern __shared__ __attribute__((aligned(16))) float4 smem4[];
    bwd_prim_body<FADE8, TS, PW, WARP, false>(p, (int)blockIdx.x, smem4);
}

// The two-pass instantiation (general slab strides): a small persistent grid that walks the same block -> primitive
// map and works only on the primitives the kernel above marked; returns at once when it marked none.  kPreciseWaves waves per
// workgroup: the marked primitives are few and LARGE (tens of thousands of samples each, marched twice), one per workgroup
// at a time, so the kernel lasts as long as its largest primitive -- more waves on it, not more workgroups, shorten that.
#ifndef MVP_PRECISE_WAVES
#define MVP_PRECISE_WAVES 4
#endif
constexpr int kPreciseWaves = MVP_PRECISE_WAVES;
template <bool FADE8, bool WARP>
__global__ __launch_bounds__(kPreciseWaves * 64, 2) void bwd_prim_precise_kernel(const MarchParams p, const int total_blocks) {
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

// ------------------------------------------------------------------------------------------------
// Grid geometry of the march kernels from (N, H, W, K): fills the fields packet_of_block / prim_of_block read.
static int setup_block_map(mvp::MarchParams &p) {
    using namespace mvp;
    p.tiles_x = (p.W + kTile - 1) / kTile;
    p.tiles_y = (p.H + kTile - 1) / kTile;
    // packet slots: an image is ceil(tiles_y / MVP_STRIP_ROWS) strips of MVP_STRIP_ROWS * tiles_x slots (packet_of_block)
    const long long strips = (p.tiles_y + MVP_STRIP_ROWS - 1) / MVP_STRIP_ROWS, S = (long long)MVP_STRIP_ROWS * p.tiles_x;
    if (strips * S > 0x3fffffffll) return MVP_ERR_UNSUPPORTED;
    p.chunk = (int)((strips * S + 7) / 8);            // 8 * chunk >= the slots of a whole image
    p.images_whole = p.N - p.N % 8;
    const int R = p.N - p.images_whole;
    // XCDs per shared image (measured, forward ms, F = 8 / 4 / 2 or 1): R = 4 (C3) 0.76 / 0.74 / 0.70, (C4) 1.00 / 0.98 /
    // 0.97; R = 2: 0.36 / 0.36 / -; R = 5: 0.61 / 0.66 / 0.80 (F = 1: three XCDs idle)
    p.band_split = R == 4 ? 2 : R == 2 ? 4 : 8;
    p.band_chunk = (int)(((strips + p.band_split - 1) / p.band_split) * S);  // slots of one XCD's share of an image
    const int rounds = (R * p.band_split + 7) / 8;  // groups of 8 / F images
    const long long blocks = 8ll * p.chunk * p.images_whole + 8ll * p.band_chunk * rounds;
    if (blocks > 0x7fffffffll) return MVP_ERR_UNSUPPORTED;
    p.total_packets = (int)blocks;
    return MVP_OK;
}

// blocks of the primitive-centric backward's grid
static long long prim_grid_blocks(const mvp::MarchParams &p) {
    return (long long)p.images_whole * p.K +
           8ll * mvp::prim_band_slots(p.K, p.band_split) * (((p.N - p.images_whole) * p.band_split + 7) / 8);
}

extern "C" int mvp_march_block_map(int N, int H, int W, int K, int kind, int first_block, int count, int *out,
                                   int *total_blocks) {
    using namespace mvp;
    if (N < 0 || H < 0 || W < 0 || K < 0 || (kind != 0 && kind != 1) || first_block < 0 || count < 0) return MVP_ERR_BADARG;
    if (count > 0 && !out) return MVP_ERR_BADARG;
    MarchParams p = {};
    p.N = N, p.H = H, p.W = W, p.K = K;
    const int rc = setup_block_map(p);
    if (rc != MVP_OK) return rc;
    const long long total = kind == 0 ? (long long)p.total_packets : prim_grid_blocks(p);
    if (total > 0x7fffffffll) return MVP_ERR_UNSUPPORTED;
    if (total_blocks) *total_blocks = (int)total;
    for (int i = 0; i < count; ++i) {
        const long long b = (long long)first_block + i;
        int n = -1, u = -1;
        const bool ok = b < total && (kind == 0 ? packet_of_block(p, (int)b, n, u) : prim_of_block(p, (int)b, n, u));
        out[2 * i] = ok ? n : -1, out[2 * i + 1] = ok ? u : -1;
    }
    return MVP_OK;
}

static int march_common_checks(bool bwd, mvp::MarchParams &p) {
    using namespace mvp;
    if (p.N < 0 || p.H < 0 || p.W < 0 || p.K < 0) return MVP_ERR_BADARG;
    if ((long long)p.N * p.H * p.W == 0) return 1;  // nothing to do
    if (!(p.stepsize > 0.f) || !(p.stepsize < INFINITY) || !(p.fadeexp > 0.f) || !(p.fadescale == p.fadescale))
        return MVP_ERR_BADARG;
    if (p.K > 0 && (p.TD < 2 || p.TH < 2 || p.TW < 2)) return MVP_ERR_UNSUPPORTED;
    if (p.K >= (1 << 24)) return MVP_ERR_UNSUPPORTED;  // list entries pack k into 24 bits
    if (p.campos) {
        if (bwd || p.raypos || p.raydir || p.tminmax) return MVP_ERR_BADARG;
        if (!p.camrot || !p.focal || !p.princpt) return MVP_ERR_BADARG;
        if (!(p.volradius > 0.f) || !(p.volradius < INFINITY)) return MVP_ERR_BADARG;
        if (p.pixelcoords && ((uintptr_t)p.pixelcoords & 7u)) return MVP_ERR_BADARG;
    } else if (!p.raypos || !p.raydir || !p.tminmax) {
        return MVP_ERR_BADARG;
    }
    if (p.K > 0 && (!p.nodeaabb || !p.primpos || !p.primrot || !p.primscale || !p.tplate)) return MVP_ERR_BADARG;
    if (!aligned16(p.tplate) || (p.tminmax && !aligned16(p.tminmax)) || !aligned16(p.nodeaabb)) return MVP_ERR_BADARG;
    if (p.pl_cap < 0 || (p.pl_cap & 3)) return MVP_ERR_BADARG;  // (lists are read four entries = 32 bytes at a time)
    if (p.rayaux && !aligned16(p.rayaux)) return MVP_ERR_BADARG;
    if (p.pl_list && !aligned16(p.pl_list)) return MVP_ERR_BADARG;
    const int rc_grid = setup_block_map(p);
    if (rc_grid != MVP_OK) return rc_grid;
#ifdef MVP_DEBUG_HOOKS
    {
        const char *e = getenv("MVP_DEBUG_FORCE_DFS");
        p.debug_force_dfs = (e && e[0] == '1') ? 1 : 0;
        const char *f = getenv("MVP_DEBUG_SLOT_SWEEP");
        p.debug_slot_sweep = (f && f[0] == '1') ? 1 : 0;
        const char *g = getenv("MVP_DEBUG_STAGE");
        p.debug_stage = g ? atoi(g) : 0;
    }
#endif
    return MVP_OK;
}

struct CameraArgs {  // mvp_march_forward_cams: rays are made inside the march
    const float *campos, *camrot, *focal, *princpt, *pixelcoords;
    float volradius;
    float *raypos_out, *raydir_out, *tminmax_out;  // all three or none
};

static int march_forward_impl(int N, int H, int W, int K, const float *raypos, const float *raydir, const CameraArgs *cams,
                              float stepsize, const float *tminmax, const float *nodeaabb, const float *primpos,
                              const float *primrot, const float *primscale, int TD, int TH, int TW,
                              const float *tplate, int WD, int WH, int WW, const float *warp, float *rayrgba,
                              float *raysat, uint32_t *rayaux, uint32_t *primlist_count, uint32_t *primlist,
                              int primlist_cap, float fadescale, float fadeexp, uint32_t *diag, void *stream) {
    using namespace mvp;
    MarchParams p = {};
    if (cams) {
        p.campos = cams->campos, p.camrot = cams->camrot, p.focal = cams->focal, p.princpt = cams->princpt;
        p.pixelcoords = cams->pixelcoords, p.volradius = cams->volradius;
        p.raypos_out = cams->raypos_out, p.raydir_out = cams->raydir_out, p.tminmax_out = cams->tminmax_out;
        if (!p.campos) return MVP_ERR_BADARG;
        const int nout = (p.raypos_out != nullptr) + (p.raydir_out != nullptr) + (p.tminmax_out != nullptr);
        if (nout != 0 && nout != 3) return MVP_ERR_BADARG;
        if (p.tminmax_out && ((uintptr_t)p.tminmax_out & 7u)) return MVP_ERR_BADARG;
    }
    p.N = N, p.H = H, p.W = W, p.K = K, p.TD = TD, p.TH = TH, p.TW = TW;
    p.WD = WD, p.WH = WH, p.WW = WW, p.warp = warp;
    if (warp && (WD < 2 || WH < 2 || WW < 2)) return MVP_ERR_UNSUPPORTED;
    p.stepsize = stepsize, p.fadescale = fadescale, p.fadeexp = fadeexp;
    p.raypos = raypos, p.raydir = raydir, p.tminmax = tminmax, p.nodeaabb = nodeaabb;
    p.primpos = primpos, p.primrot = primrot, p.primscale = primscale, p.tplate = tplate;
    p.rayrgba = rayrgba, p.raysat = raysat, p.diag = diag;
    p.rayaux = rayaux, p.pl_count = primlist_count, p.pl_list = reinterpret_cast<uint2 *>(primlist);
    p.pl_cap = primlist_cap;
    int rc = march_common_checks(false, p);
    if (rc == 1) return MVP_OK;
    if (rc != MVP_OK) return rc;
    if (!rayrgba || !aligned16(rayrgba)) return MVP_ERR_BADARG;
    if ((primlist_count != nullptr) != (primlist != nullptr)) return MVP_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    if (K == 0) {  // nothing to march through: all-zero image; raysat would need a -1 fill (never on the training path)
        if (raysat) return MVP_ERR_UNSUPPORTED;
        hipError_t e = hipMemsetAsync(rayrgba, 0, sizeof(float) * 4 * (size_t)N * H * W, st);
        return e == hipSuccess ? MVP_OK : (int)e;
    }
    if (p.pl_count) {
        if ((long long)p.tiles_x * p.tiles_y > (1ll << 23)) {  // packet index does not fit the packed list entry
            p.pl_count = nullptr, p.pl_list = nullptr;        // backward will see the global flag set below
        }
        hipError_t e = hipMemsetAsync(primlist_count, 0,
                                      sizeof(uint32_t) * ((size_t)N * K + 3 + (size_t)N * p.tiles_x * p.tiles_y), st);
        if (e != hipSuccess) return (int)e;
        if (!p.pl_count) {
            e = hipMemsetD32Async((hipDeviceptr_t)(primlist_count + (size_t)N * K), (int)kFlagGlobal, 1, st);
            if (e != hipSuccess) return (int)e;
        }
        // tail[2] = bits(1.0f): max |raysat| of an image in which no ray saturates (raysat = -1)
        e = hipMemsetD32Async((hipDeviceptr_t)(primlist_count + (size_t)N * K + 2), 0x3f800000, 1, st);
        if (e != hipSuccess) return (int)e;
    }
    const bool fade8 = fadeexp == 8.0f;
#if MVP_FWD_QUAD > 1
    const dim3 grid((unsigned)(8 * (((p.total_packets >> 3) + MVP_FWD_QUAD - 1) / MVP_FWD_QUAD))), block(kWave * MVP_FWD_QUAD);
#else
    const dim3 grid((unsigned)p.total_packets), block(kWave);
#endif
    if (warp) {
        if (fade8)
            hipLaunchKernelGGL((march_kernel<false, true, true>), grid, block, 0, st, p);
        else
            hipLaunchKernelGGL((march_kernel<false, false, true>), grid, block, 0, st, p);
    } else {
        // the reference's slab size (and BASELINE's) gets compile-time strides and 32-bit slab offsets
        const bool cube8 = TD == 8 && TH == 8 && TW == 8 && (unsigned long long)K * 8192ull < (1ull << 32);
        if (fade8 && cube8)
            hipLaunchKernelGGL((march_kernel<false, true, false, 8>), grid, block, 0, st, p);
        else if (fade8)
            hipLaunchKernelGGL((march_kernel<false, true, false>), grid, block, 0, st, p);
        else if (cube8)
            hipLaunchKernelGGL((march_kernel<false, false, false, 8>), grid, block, 0, st, p);
        else
            hipLaunchKernelGGL((march_kernel<false, false, false>), grid, block, 0, st, p);
    }
    return launch_status();
}

extern "C" int mvp_march_forward(int N, int H, int W, int K, const float *raypos, const float *raydir,
                                 float stepsize, const float *tminmax, const float *nodeaabb, const float *primpos,
                                 const float *primrot, const float *primscale, int TD, int TH, int TW,
                                 const float *tplate, int WD, int WH, int WW, const float *warp, float *rayrgba,
                                 float *raysat, uint32_t *rayaux, uint32_t *primlist_count, uint32_t *primlist,
                                 int primlist_cap, float fadescale, float fadeexp, uint32_t *diag, void *stream) {
    return march_forward_impl(N, H, W, K, raypos, raydir, nullptr, stepsize, tminmax, nodeaabb, primpos, primrot,
                              primscale, TD, TH, TW, tplate, WD, WH, WW, warp, rayrgba, raysat, rayaux, primlist_count,
                              primlist, primlist_cap, fadescale, fadeexp, diag, stream);
}

extern "C" int mvp_march_forward_cams(int N, int H, int W, int K, const float *campos, const float *camrot,
                                      const float *focal, const float *princpt, const float *pixelcoords,
                                      float volradius, float stepsize, const float *nodeaabb, const float *primpos,
                                      const float *primrot, const float *primscale, int TD, int TH, int TW,
                                      const float *tplate, float *rayrgba, float *raysat, uint32_t *rayaux,
                                      uint32_t *primlist_count, uint32_t *primlist, int primlist_cap,
                                      float *raypos_out, float *raydir_out, float *tminmax_out, float fadescale,
                                      float fadeexp, uint32_t *diag, void *stream) {
    const CameraArgs cams = {campos, camrot, focal, princpt, pixelcoords, volradius, raypos_out, raydir_out, tminmax_out};
    return march_forward_impl(N, H, W, K, nullptr, nullptr, &cams, stepsize, nullptr, nodeaabb, primpos, primrot,
                              primscale, TD, TH, TW, tplate, 0, 0, 0, nullptr, rayrgba, raysat, rayaux, primlist_count,
                              primlist, primlist_cap, fadescale, fadeexp, diag, stream);
}

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
    p.pl_list = reinterpret_cast<uint2 *>(const_cast<uint32_t *>(primlist)), p.pl_cap = primlist_cap;
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
    const size_t Vp = (size_t)TD * ((size_t)TH * TW + kGradPadZ);
    // float4 slab + [4][Vp] int32 + ray queue + reduce area (+ queue tail)
    // 2 or 3 waves per workgroup by the work a primitive has: ray packets per primitive (see the note at kEntriesPerWave)
    const int pw = ((long long)p.tiles_x * p.tiles_y * 4 > 5ll * K) ? 3 : 2;
    // + list indices by rank (2 bytes per list slot; used by primitives whose list needs more than one round)
    size_t lds = V * 16 + Vp * 16 + (size_t)prim_queue_cap(pw) * 8 + 64 * sizeof(float) + 16 + kLenBuckets * 4 +
                 (((size_t)(primlist_cap > 0 ? primlist_cap : 0) * 2 + 15) & ~(size_t)15);
    if (warp) {  // + the warp grid (float4 per node) and its [3][VWp] accumulators
        lds = (lds + 15) & ~(size_t)15;
        p.prim_lds_base = (int)lds;
        const size_t VW = (size_t)WD * WH * WW, VWp = (size_t)WD * ((size_t)WH * WW + kGradPadZ);
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
                const size_t VW = (size_t)WD * WH * WW, VWp = (size_t)WD * ((size_t)WH * WW + kGradPadZ);
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
