// aabb.hip -- AABBs of the implicit heap BVH ("fixedorder" tree) for gfx950.
// Semantics: /root/reference/extensions/mvpraymarch/bvh.cu:157-201 (leaf box + bottom-up union) with the leaf
// formula of primtransf.h:12-63 and the heap topology of mvpraymarch.py:57-75.
//
// The reference walks leaf->root with atomicCAS arrival flags in a scratch buffer it cudaMalloc's, memsets and
// frees on every call (bvh.cu:261-263,293).  Because the fixed-order tree is an implicit heap, no flags, no
// topology tensors and no scratch are needed here: the tree is cut into tiers of kTierLevels levels; one
// workgroup reduces one sub-heap of a tier entirely in LDS (leaves computed in place, parents = union of the
// two LDS children) and writes every node once.  Tiers run bottom-up as successive launches on the stream
// (2 launches up to K = 2^15 leaves), so the only inter-workgroup hand-off is a kernel boundary.
#include "mvp_device.h"
#include "mvp_host.h"

namespace mvp {

constexpr int kTierLevels = 8;                       // levels per tier -> 255 LDS nodes (6 KB) per workgroup
constexpr int kTierNodes = (1 << kTierLevels) - 1;
constexpr int kAabbBlock = 128;

struct Box {
    float lo[3], hi[3];
};

// primtransf.h:12-63: corner c in {-1,1}^3, p = c / scale, world_i = dot(p, R_i) + pos_i (R_i = row i)
__device__ __forceinline__ Box leaf_box(const float *__restrict__ pos, const float *__restrict__ rot,
                                        const float *__restrict__ scale) {
    const f3 t = ld3(pos), r0 = ld3(rot), r1 = ld3(rot + 3), r2 = ld3(rot + 6), s = ld3(scale);
    const f3 e = mk3(1.0f / s.x, 1.0f / s.y, 1.0f / s.z);
    Box b;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f3 p = mk3((c & 1) ? e.x : -e.x, (c & 2) ? e.y : -e.y, (c & 4) ? e.z : -e.z);
        const float x = dot3(p, r0) + t.x, y = dot3(p, r1) + t.y, z = dot3(p, r2) + t.z;
        if (c == 0) {
            b.lo[0] = b.hi[0] = x;
            b.lo[1] = b.hi[1] = y;
            b.lo[2] = b.hi[2] = z;
        } else {
            b.lo[0] = fminf(b.lo[0], x), b.hi[0] = fmaxf(b.hi[0], x);
            b.lo[1] = fminf(b.lo[1], y), b.hi[1] = fmaxf(b.hi[1], y);
            b.lo[2] = fminf(b.lo[2], z), b.hi[2] = fmaxf(b.hi[2], z);
        }
    }
    return b;
}

__device__ __forceinline__ Box box_union(const float *a, const float *b) {
    Box o;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        o.lo[j] = fminf(a[j], b[j]);
        o.hi[j] = fmaxf(a[3 + j], b[3 + j]);
    }
    return o;
}

// One workgroup = one sub-heap rooted at global node `root` (depth dlo), levels dlo..dhi, of image n.
__global__ __launch_bounds__(kAabbBlock) void aabb_tier_kernel(int K, int dlo, int dhi,
                                                               const float *__restrict__ primpos,
                                                               const float *__restrict__ primrot,
                                                               const float *__restrict__ primscale,
                                                               float *__restrict__ nodeaabb) {
    __shared__ float s_box[kTierNodes * 6];
    const int NN = 2 * K - 1;
    const int n = blockIdx.y;
    const int root = (1 << dlo) - 1 + (int)blockIdx.x;
    if (root >= NN) return;
    const int L = dhi - dlo + 1;
    const float *pp = primpos + (size_t)n * K * 3;
    const float *pr = primrot + (size_t)n * K * 9;
    const float *ps = primscale + (size_t)n * K * 3;
    float *A = nodeaabb + (size_t)n * NN * 6;
    for (int l = L - 1; l >= 0; --l) {
        const int cnt = 1 << l;
        const int gfirst = ((root + 1) << l) - 1;  // leftmost descendant of `root` at relative level l
        for (int j = threadIdx.x; j < cnt; j += kAabbBlock) {
            const int g = gfirst + j;
            if (g >= NN) break;
            const int li = cnt - 1 + j;  // local heap index
            Box b;
            if (g >= K - 1) {  // leaf: node K-1+k is primitive k (mvpraymarch.py:45)
                const int k = g - (K - 1);
                b = leaf_box(pp + (size_t)k * 3, pr + (size_t)k * 9, ps + (size_t)k * 3);
            } else if (l == L - 1) {  // internal node on the tier floor: children were written by the tier below
                b = box_union(A + (size_t)(2 * g + 1) * 6, A + (size_t)(2 * g + 2) * 6);
            } else {  // children live in LDS
                b = box_union(s_box + (2 * li + 1) * 6, s_box + (2 * li + 2) * 6);
            }
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                s_box[li * 6 + q] = b.lo[q];
                s_box[li * 6 + 3 + q] = b.hi[q];
                A[(size_t)g * 6 + q] = b.lo[q];
                A[(size_t)g * 6 + 3 + q] = b.hi[q];
            }
        }
        __syncthreads();
    }
}

inline int ilog2_floor(unsigned v) {
    int r = -1;
    while (v) {
        ++r;
        v >>= 1;
    }
    return r;
}

}  // namespace mvp

extern "C" int mvp_aabb_build(int N, int K, const float *primpos, const float *primrot, const float *primscale,
                              float *nodeaabb, void *stream) {
    if (N < 0 || K < 0) return MVP_ERR_BADARG;
    if (N == 0 || K == 0) return MVP_OK;
    if (!primpos || !primrot || !primscale || !nodeaabb) return MVP_ERR_BADARG;
    if (K > (1 << 28)) return MVP_ERR_UNSUPPORTED;
    if (N > 65535) return MVP_ERR_UNSUPPORTED;  // grid.y
    const int NN = 2 * K - 1;
    const int dmax = mvp::ilog2_floor((unsigned)NN);  // depth of the last node; depth(i) = floor(log2(i+1))
    int dhi = dmax;
    while (dhi >= 0) {
        int dlo = dhi - (mvp::kTierLevels - 1);
        if (dlo < 0) dlo = 0;
        long long roots = 1ll << dlo;
        const long long avail = (long long)NN - ((1ll << dlo) - 1);
        if (roots > avail) roots = avail;
        hipLaunchKernelGGL(mvp::aabb_tier_kernel, dim3((unsigned)roots, (unsigned)N), dim3(mvp::kAabbBlock), 0,
                           (hipStream_t)stream, K, dlo, dhi, primpos, primrot, primscale, nodeaabb);
        int rc = mvp::launch_status();
        if (rc != MVP_OK) return rc;
        dhi = dlo - 1;
    }
    return MVP_OK;
}
