// march_packet.h -- one 8x8 ray packet marched by one wave64: the forward kernel (march_kernel<false, ..>) and the
// ray-centric backward (march_kernel<true, ..>: global fp32 atomics; the always-correct owner of whatever the
// primitive-centric backward of march_bwd.hip hands over).  Schedule and reference citations: march_common.h.
#pragma once
#include "march_common.h"

namespace mvp {

// LDS ordering point inside a packet.  One wave = one workgroup: a workgroup barrier, which is free there.  (Workgroups of
// several packets -- fences instead of barriers, so that neighbouring packets share a CU's L1 -- were measured in round 4:
// profiles/r04_fwd_ab_quad.txt, profiles/r05_timing_variants.patch.)
__device__ __forceinline__ void packet_sync() { __syncthreads(); }

// HALF (forward, TS > 0, no warp field, no hand-off): p.tplate points at fp16 RGBA slabs (sample_slab_h)
template <bool BWD, bool FADE8, bool WARP, int TS, bool HALF = false>
__device__ __forceinline__ void march_packet(const MarchParams &p, const int b, int *s_a, int *s_b, float4 *s_rec,
                                             uint32_t *s_tab, const bool emit_all) {
    constexpr bool FAST = !BWD;  // the lane-independent sweep exists for the forward only (round 6: the warp-field one too)
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

    // Every ray against the ROOT box first (utils.h:679-685 with the packet test's slack; a primitive a ray can sample lies
    // inside it): more than half of a head shot's packets see only background, and this is ~25 instructions against the ~130 of
    // the packet bounds (three divisions, eight wave reductions) they would otherwise compute before the packet-level root test
    // tells them the same (round 6: C2 forward 5.270 -> 5.236 ms, C4 -0.5 %, C3 +-0: profiles/r06_fwd_ab_root_per_ray.txt).
    bool root_any = false;
    if (__ballot(active) != 0ull) {
        const float2 *ap = reinterpret_cast<const float2 *>(A);  // root AABB, wave-uniform
        const float2 a0 = ap[0], a1 = ap[1], a2 = ap[2];
        const f3 ird = mk3(fast_rcp(d.x), fast_rcp(d.y), fast_rcp(d.z));
        const f3 t0 = mk3((a0.x - o.x) * ird.x, (a0.y - o.y) * ird.y, (a1.x - o.z) * ird.z);
        const f3 t1 = mk3((a1.y - o.x) * ird.x, (a2.x - o.y) * ird.y, (a2.y - o.z) * ird.z);
        // (a NaN bound or quotient drops out of the min / max, as in packet_hits_box; an all-NaN box fails the comparison)
        const float tn = fmaxf(max3f(fminf(t0.x, t1.x), fminf(t0.y, t1.y), fminf(t0.z, t1.z)), tmin);
        const float tf = fminf(min3f(fmaxf(t0.x, t1.x), fmaxf(t0.y, t1.y), fmaxf(t0.z, t1.z)), tmax + 1e-5f);
        root_any = __ballot(active && tn <= tf + 1e-4f + 1e-5f * fabsf(tf)) != 0ull;
    }
    if (root_any) {
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
    unsigned long long msk0 = 0ull;  // ... and which rays of the packet have a step in it (the backward's phase 1 examines only those)
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
            const unsigned long long somem = __ballot(some);
            if (somem != 0ull) {  // wave-uniform
                if constexpr (HALF) {
                    // render path: no list is handed to a backward, so the packet's step range (two wave reductions per listed
                    // primitive) is not needed -- only the test that every step index fits the crossing table's field
                    if (nh >= kFastSlots || __ballot(some && hi > kFastMaxStep) != 0ull) {
                        wfail = true;
                        break;
                    }
                } else {
                    const int wlo = uni(wave_min(lo)), whi = uni(wave_max(hi));
                    if (nh >= kFastSlots || whi > kFastMaxStep) {
                        wfail = true;
                        break;
                    }
                    if (lane == nh) {
                        ent0 = k;
                        rg0 = wlo | (whi << 16);
                        msk0 = somem;
                    }
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
        // (HALF: the same pointer addresses fp16 RGBA slabs, 8 bytes per voxel)
        const char *Th = reinterpret_cast<const char *>(p.tplate) + (size_t)n * K * (V4 * 2);
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
                            if constexpr (WARP)
                                v = sample_warped<FADE8>(p.warp + ((size_t)n * K + k) * ((size_t)p.WD * p.WH * p.WW * 3),
                                                         T + (size_t)k * V4, y, p.WD, p.WH, p.WW, p.TD, p.TH, p.TW,
                                                         p.fadescale, p.fadeexp);
                            else if constexpr (HALF)
                                v = sample_slab_h<FADE8, TS>(Th, (uint32_t)k * (uint32_t)(TS * TS * TS * 8), y, p.fadescale,
                                                             p.fadeexp);
                            else if constexpr (TS > 0)
                                v = sample_slab_c<FADE8, TS>(T, (uint32_t)k * (uint32_t)(TS * TS * TS * 16), y, p.fadescale,
                                                             p.fadeexp);
                            else
                                v = sample_slab<FADE8>(T + (size_t)k * V4, y, p.TD, p.TH, p.TW, p.fadescale, p.fadeexp);
                            float contrib;
                            if constexpr (!HALF) nanw = nanw || (v.w != v.w);   // (only the hand-off to a backward reads it)
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
                                    v = sample_warped<FADE8>(p.warp + ((size_t)n * K + k) * VW3, T + (size_t)k * V4, y, p.WD,
                                                             p.WH, p.WW, p.TD, p.TH, p.TW, p.fadescale, p.fadeexp);
                                } else {
                                    if constexpr (HALF)
                                        v = sample_slab_h<FADE8, TS>(Th, (uint32_t)k * (uint32_t)(TS * TS * TS * 8), y,
                                                                     p.fadescale, p.fadeexp);
                                    else if constexpr (TS > 0)
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

    // ---------------- grad mode: hand this packet's list to the primitive-centric backward ----------------
    // BEHIND the sweep (round 6; by itself a measured zero, profiles/r06_fwd_append_after.txt): the sweep knows where every ray
    // SATURATED, and nothing behind that sample was evaluated (primaccum.h:63-79, subset_kernel.h:76) -- on a trained-like scene
    // the rays saturate in the front shell and about half of the (ray, primitive) pairs the exact test listed lie entirely behind
    // that point.  In lane-independent mode every saturated ray clears its bit in the masks of the list slots whose FIRST sample
    // of that ray comes after its saturation key (its crossing table is still in LDS; the record area, free now, holds the masks:
    // one 64-bit LDS atomic per dead crossing, nothing at all in packets without a saturated ray), and a record whose mask is
    // empty is not appended: the backward's phase 1 neither loads it nor examines its rays.  The entries themselves are where the
    // exact test left them: (ent0, rg0, msk0) in registers, s_b / s_a (which the slot-synchronous sweep only reads) otherwise.
    if (!BWD && !HALF && p.pl_count != nullptr && nh > 0) {
        uint32_t *flags = p.pl_count + (size_t)p.N * K;
        if (FAST && fast) {
            const bool satd = satkey != kNoSat;
            if (__ballot(satd) != 0ull) {  // (wave-uniform)
                unsigned long long *s_mask = reinterpret_cast<unsigned long long *>(s_rec);
                packet_sync();
                if (lane < nh) s_mask[lane] = msk0;
                packet_sync();
                if (satd) {
                    for (int c = 0; c < ncross; ++c) {  // this ray's crossings, in list order
                        const uint32_t e = s_tab[c * kWave + lane];
                        const uint32_t slot = e & 63u;
                        if ((((e >> 17) << 9) | slot) > satkey) atomicAnd(s_mask + slot, ~(1ull << lane));
                    }
                }
                packet_sync();
                if (lane < nh) msk0 = s_mask[lane];
            }
            if (lane < nh && msk0 != 0ull) {
                const size_t pk = (size_t)n * K + ent0;
                const uint32_t idx = atomicAdd(p.pl_count + pk, 1u);
                if (idx < (uint32_t)p.pl_cap) {
                    p.pl_list[pk * (size_t)p.pl_cap + idx] = make_uint4(((uint32_t)tidx << 9) | (uint32_t)lane, (uint32_t)rg0,
                                                                        (uint32_t)msk0, (uint32_t)(msk0 >> 32));
                } else {
                    raise_flag(flags, kFlagListOverflow);
                    flags[3 + (size_t)n * p.tiles_x * p.tiles_y + tidx] = kPacketFwdOverflow;  // (region zeroed by the host)
                }
            }
        } else {
            // (the slot-synchronous layout has no room for a mask per list slot -- up to 512 of them: every active ray is a
            //  candidate for every entry of such a packet, as it was for all packets before round 6)
            const unsigned long long actm = __ballot(active);
            for (int j = lane; j < nh; j += kWave) {
                const int k = s_b[j] & 0xffffff;
                const size_t pk = (size_t)n * K + k;
                const uint32_t idx = atomicAdd(p.pl_count + pk, 1u);
                if (idx < (uint32_t)p.pl_cap) {
                    p.pl_list[pk * (size_t)p.pl_cap + idx] = make_uint4(((uint32_t)tidx << 9) | (uint32_t)j, (uint32_t)s_a[j],
                                                                        (uint32_t)actm, (uint32_t)(actm >> 32));
                } else {
                    raise_flag(flags, kFlagListOverflow);
                    flags[3 + (size_t)n * p.tiles_x * p.tiles_y + tidx] = kPacketFwdOverflow;
                }
            }
            if (!ranges_ok && lane == 0) raise_flag(flags, kFlagGlobal);
        }
    }

    if (!BWD && !HALF && p.pl_count != nullptr) {
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
        if constexpr (HALF) return;   // (render path: nothing is handed to a backward)
        if (p.raysat) {
            float *sp = p.raysat + r * 3;
            MVP_STREAM_STOREF(sp, raysat.x), MVP_STREAM_STOREF(sp + 1, raysat.y), MVP_STREAM_STOREF(sp + 2, raysat.z);
        }
        // (the backward reads a ray's record only through a list entry that names the ray: a packet that lists no primitive
        //  -- more than half of them -- has no reader for its 16 bytes per ray)
        if (p.rayaux && nh > 0)
            MVP_STREAM_STORE(reinterpret_cast<float4 *>(p.rayaux) + r,
                             make_float4(__uint_as_float(satkey), wbefore, __uint_as_float((uint32_t)incs), tend));
    }
}

// TS > 0 (forward, no warp field): TS^3 slabs with compile-time strides, see sample_slab_c
template <bool BWD, bool FADE8, bool WARP, int TS = 0>
__global__ __launch_bounds__(kWave) void march_kernel(const MarchParams p) {
    // One LDS block per wave: [SRT records: 64 x 64 B][region].  The region is the two 512-entry frontier / list arrays
    // (s_a, s_b); in the plain forward it is large enough to be re-used, after the traversal, as the per-ray crossing
    // table of the lane-independent sweep (kFastCross rows x 64 lanes x 4 B).
    // slot-synchronous layout: 64 records + s_a + s_b; lane-independent layout: kFastSlots records + kFastCross rows
    constexpr int kSlowWords = kRecSlots * 16 + 2 * kMaxList;
    constexpr int kFastWords = kFastSlots * 16 + kFastCross * kWave;
    constexpr int kWords = (!BWD && kFastWords > kSlowWords) ? kFastWords : kSlowWords;
    __shared__ __attribute__((aligned(16))) uint32_t smem[kWords];
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
    } else {
        march_packet<BWD, FADE8, WARP, TS>(p, blockIdx.x, s_a, s_b, s_rec, s_tab, true);
    }
}

// The opt-in render path over fp16 slabs (8^3 only): forward without hand-off, see sample_slab_h.
template <bool FADE8>
__global__ __launch_bounds__(kWave) void march_half_kernel(const MarchParams p) {
    constexpr int kSlowWords = kRecSlots * 16 + 2 * kMaxList;
    constexpr int kFastWords = kFastSlots * 16 + kFastCross * kWave;
    constexpr int kWords = kFastWords > kSlowWords ? kFastWords : kSlowWords;
    __shared__ __attribute__((aligned(16))) uint32_t smem[kWords];
    float4 *s_rec = reinterpret_cast<float4 *>(smem);
    int *s_a = reinterpret_cast<int *>(smem + kRecSlots * 16);
    int *s_b = s_a + kMaxList;
    uint32_t *s_tab = smem + kFastSlots * 16;
    march_packet<false, FADE8, false, 8, true>(p, blockIdx.x, s_a, s_b, s_rec, s_tab, true);
}

}  // namespace mvp
