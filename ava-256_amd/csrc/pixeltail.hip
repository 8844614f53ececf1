// pixeltail.hip -- the decode tail of the autoencoder and its image loss as ONE pass each way.
//
// What it replaces in the reference (all eager PyTorch, ~10 full-image kernels forward and ~20 backward, plus the
// Raymarcher's permute + two .contiguous() copies, models/raymarchers/mvpraymarcher.py:50-51):
//   models/autoencoder.py:254-256    rayrgb = colorcal(rayrgb, camindex, idindex)
//   models/colorcals/colorcal.py:28-31   w = wcam[cam] + wident[id]; b = bcam[cam] + bident[id]; w * image + b
//   models/autoencoder.py:263-265    rayrgb = rayrgb + (1 - rayalpha) * bg
//   losses.py:12-14 / ddp-train.py:404-405   irgbl1 = mean(|irgbrec - image|)
// Forward: reads the march's own output layout rayrgba [N,H,W,4] (no NHWC -> NCHW split pass), the per-image colour affine
// (w, b: the two tiny index-adds stay in PyTorch), the background planes and the target image; writes irgbrec [N,3,H,W],
// ialpha [N,1,H,W] and per-workgroup partial sums of |irgbrec - image|.  Backward: reads the upstream gradient of irgbrec
// (optional) and of the L1 sum (a device scalar), writes grad_rayrgba [N,H,W,4] -- the layout the march backward takes --
// grad_bg planes and per-workgroup partial sums of the colour-affine gradients.
// Arithmetic: the reference's operations in the reference's order, each rounded once (fp contraction off), so irgbrec is
// bit-identical to the eager expression.  One thread = one pixel: a 16-byte RGBA access and 4-byte plane accesses, all
// coalesced.  HBM-bound: 56 B per pixel forward, 92 B backward (with every optional tensor present).
#include "mvp_device.h"
#include "mvp_host.h"

namespace mvp {

constexpr int kTailBlock = 256;

__device__ __forceinline__ float block_sum(float v, float *s_part) {  // result valid in thread 0
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_part[wave] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kTailBlock / 64; ++w) r += s_part[w];
    }
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(kTailBlock) void pixel_tail_fwd_kernel(int HW, const float4 *__restrict__ rgba,
                                                                    const float *__restrict__ cw, const float *__restrict__ cb,
                                                                    const float *__restrict__ bg, const float *__restrict__ target,
                                                                    float *__restrict__ irgbrec, float *__restrict__ ialpha,
                                                                    float *__restrict__ l1_partials) {
#pragma clang fp contract(off)
    __shared__ float s_part[kTailBlock / 64];
    const int n = blockIdx.y;
    const int p = blockIdx.x * kTailBlock + threadIdx.x;
    float acc = 0.f;
    if (p < HW) {
        const float4 v = rgba[(size_t)n * HW + p];
        const size_t o3 = (size_t)n * 3 * HW + p;
        float c[3] = {v.x, v.y, v.z};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (cw) c[j] = cw[n * 3 + j] * c[j] + cb[n * 3 + j];            // colorcal.py:31
            if (bg) c[j] = c[j] + (1.0f - v.w) * bg[o3 + (size_t)j * HW];   // autoencoder.py:264
            irgbrec[o3 + (size_t)j * HW] = c[j];
            if (target) acc += fabsf(c[j] - target[o3 + (size_t)j * HW]);   // losses.py:13-14
        }
        ialpha[(size_t)n * HW + p] = v.w;
    }
    if (l1_partials) {
        const float s = block_sum(acc, s_part);
        if (threadIdx.x == 0) l1_partials[(size_t)n * gridDim.x + blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(kTailBlock) void pixel_tail_bwd_kernel(int HW, const float4 *__restrict__ rgba,
                                                                    const float *__restrict__ cw, const float *__restrict__ bg,
                                                                    const float *__restrict__ target,
                                                                    const float *__restrict__ irgbrec,
                                                                    const float *__restrict__ g_irgbrec,
                                                                    const float *__restrict__ g_ialpha,
                                                                    const float *__restrict__ g_l1,
                                                                    float4 *__restrict__ grad_rgba, float *__restrict__ grad_bg,
                                                                    float *__restrict__ cwcb_partials) {
    __shared__ float s_part[kTailBlock / 64];
    const int n = blockIdx.y;
    const int p = blockIdx.x * kTailBlock + threadIdx.x;
    const float gl = (g_l1 && target) ? cload(g_l1) : 0.f;
    float sw[3] = {0.f, 0.f, 0.f}, sb[3] = {0.f, 0.f, 0.f};
    if (p < HW) {
        const float4 v = rgba[(size_t)n * HW + p];
        const size_t o3 = (size_t)n * 3 * HW + p;
        const float c[3] = {v.x, v.y, v.z};
        float4 g = make_float4(0.f, 0.f, 0.f, g_ialpha ? g_ialpha[(size_t)n * HW + p] : 0.f);
        float ga[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float G = g_irgbrec ? g_irgbrec[o3 + (size_t)j * HW] : 0.f;
            if (target) {  // d|x|/dx = sign(x), 0 at 0 (torch.abs)
                const float d = irgbrec[o3 + (size_t)j * HW] - target[o3 + (size_t)j * HW];
                G += gl * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
            }
            if (bg) {
                const float b = bg[o3 + (size_t)j * HW];
                g.w -= G * b;
                grad_bg[o3 + (size_t)j * HW] = G * (1.0f - v.w);
            }
            ga[j] = cw ? G * cw[n * 3 + j] : G;
            sw[j] = G * c[j], sb[j] = G;
        }
        g.x = ga[0], g.y = ga[1], g.z = ga[2];
        grad_rgba[(size_t)n * HW + p] = g;
    }
    if (cwcb_partials) {
        float *out = cwcb_partials + ((size_t)n * gridDim.x + blockIdx.x) * 6;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float a = block_sum(sw[j], s_part), b = block_sum(sb[j], s_part);
            if (threadIdx.x == 0) out[j] = a, out[3 + j] = b;
        }
    }
}

}  // namespace mvp

extern "C" int mvp_pixel_tail_blocks(int H, int W) {
    const long long hw = (long long)H * W;
    return hw <= 0 ? 0 : (int)((hw + mvp::kTailBlock - 1) / mvp::kTailBlock);
}

extern "C" int mvp_pixel_tail_forward(int N, int H, int W, const float *rayrgba, const float *cw, const float *cb,
                                      const float *bg, const float *target, float *irgbrec, float *ialpha,
                                      float *l1_partials, void *stream) {
    if (N < 0 || H < 0 || W < 0) return MVP_ERR_BADARG;
    const long long hw = (long long)H * W;
    if (N == 0 || hw == 0) return MVP_OK;
    if (!rayrgba || !irgbrec || !ialpha || !mvp::aligned16(rayrgba)) return MVP_ERR_BADARG;
    if ((cw == nullptr) != (cb == nullptr)) return MVP_ERR_BADARG;
    if (l1_partials && !target) return MVP_ERR_BADARG;
    if (hw > 0x7fffffffll || N > 65535) return MVP_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)mvp_pixel_tail_blocks(H, W), (unsigned)N);
    hipLaunchKernelGGL(mvp::pixel_tail_fwd_kernel, grid, dim3(mvp::kTailBlock), 0, (hipStream_t)stream, (int)hw,
                       reinterpret_cast<const float4 *>(rayrgba), cw, cb, bg, target, irgbrec, ialpha, l1_partials);
    return mvp::launch_status();
}

extern "C" int mvp_pixel_tail_backward(int N, int H, int W, const float *rayrgba, const float *cw, const float *bg,
                                       const float *target, const float *irgbrec, const float *g_irgbrec,
                                       const float *g_ialpha, const float *g_l1, float *grad_rayrgba, float *grad_bg,
                                       float *cwcb_partials, void *stream) {
    if (N < 0 || H < 0 || W < 0) return MVP_ERR_BADARG;
    const long long hw = (long long)H * W;
    if (N == 0 || hw == 0) return MVP_OK;
    if (!rayrgba || !grad_rayrgba || !mvp::aligned16(rayrgba) || !mvp::aligned16(grad_rayrgba)) return MVP_ERR_BADARG;
    if (target && !irgbrec) return MVP_ERR_BADARG;
    if ((bg == nullptr) != (grad_bg == nullptr)) return MVP_ERR_BADARG;
    if (cwcb_partials && !cw) return MVP_ERR_BADARG;
    if (hw > 0x7fffffffll || N > 65535) return MVP_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)mvp_pixel_tail_blocks(H, W), (unsigned)N);
    hipLaunchKernelGGL(mvp::pixel_tail_bwd_kernel, grid, dim3(mvp::kTailBlock), 0, (hipStream_t)stream, (int)hw,
                       reinterpret_cast<const float4 *>(rayrgba), cw, bg, target, irgbrec, g_irgbrec, g_ialpha, g_l1,
                       reinterpret_cast<float4 *>(grad_rayrgba), grad_bg, cwcb_partials);
    return mvp::launch_status();
}
