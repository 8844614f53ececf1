// assemble.hip -- decoder -> raymarch hand-off: builds the channels-last RGBA slab tensor the march reads directly
// from the two conv-decoder outputs, in one pass.
//
// What it replaces in the reference (three separate places, ~4 full passes over a 134 MB/image tensor in eager PyTorch):
//   models/decoders/rgb.py:137-143        rgb.view(N, B, 3, h, B, w, B).permute(0,3,5,1,4,6,2).reshape(N, h*w, B,B,B, 3)
//   models/decoders/geometry.py:183-185   the same with 1 channel for the opacity
//   models/decoders/assembler.py:261      template = cat([relu(rgb * 25 + 100), relu(alpha)], dim=-1)
// i.e.  tplate[n, hy*nh + wx, z, y, x, c] = relu(tex[n, z*3 + c, hy*B + y, wx*B + x] * 25 + 100)   (c < 3)
//       tplate[n, hy*nh + wx, z, y, x, 3] = relu(opacity[n, z, hy*B + y, wx*B + x])
// Pure data movement: HBM-bound, 32 B read + 32 B written per voxel... per 2 voxels: 16 B/voxel each way.
//
// Mapping for CDNA4: a thread owns 4 consecutive x of one (n, z, image row): four 16-byte loads (three tex planes and
// the opacity plane; a wave reads 1 KiB contiguous per plane) and four 16-byte stores that form 64 contiguous bytes
// of the slab row; two neighbouring lanes complete a 128-byte line.  No LDS, no atomics.
// The products use separate multiply and add (no FMA contraction) so that the result is bit-identical to the eager
// PyTorch expression.
#include "mvp_device.h"
#include "mvp_host.h"

namespace mvp {

__device__ __forceinline__ float relu_keep_nan(float v) { return v <= 0.f ? 0.f : v; }  // NaN stays NaN like torch.relu

// rgb * 25 + 100 with TWO roundings, as the eager expression computes it (hipcc contracts a*b+c to an FMA by default,
// also through __fmul_rn/__fadd_rn, which are plain operators in HIP)
__device__ __forceinline__ float rgb_denorm(float v) {
#pragma clang fp contract(off)
    const float m = v * 25.0f;
    return m + 100.0f;
}

__global__ __launch_bounds__(256) void assemble_fwd_kernel(int N, int nh, int B, const float *__restrict__ tex,
                                                           const float *__restrict__ opac,
                                                           float *__restrict__ tplate) {
    // grid = (x chunks of one image row, image rows, N * B): no 64-bit index arithmetic (three 64-bit div/mod per
    // 128 bytes moved made the first version of these kernels instruction-bound)
    const int S = nh * B, S4 = S >> 2;
    const int X4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (X4 < S4) {
        const int R = blockIdx.y;
        const int n = (int)blockIdx.z / B, z = (int)blockIdx.z - n * B;
        const int X = X4 << 2;
        const size_t plane = (size_t)S * S, rowoff = (size_t)R * S + X;
        const float *tp = tex + ((size_t)n * 3 * B + (size_t)z * 3) * plane + rowoff;
        const float4 r4 = *reinterpret_cast<const float4 *>(tp);
        const float4 g4 = *reinterpret_cast<const float4 *>(tp + plane);
        const float4 b4 = *reinterpret_cast<const float4 *>(tp + 2 * plane);
        const float4 a4 = *reinterpret_cast<const float4 *>(opac + ((size_t)n * B + z) * plane + rowoff);
        const int hy = R / B, y = R - hy * B, wx = X / B, x = X - wx * B;
        float4 *out = reinterpret_cast<float4 *>(tplate) +
                      ((((size_t)n * nh * nh + (size_t)hy * nh + wx) * B + z) * B + y) * B + x;
#define MVP_RGB(V_) relu_keep_nan(rgb_denorm(V_))
        out[0] = make_float4(MVP_RGB(r4.x), MVP_RGB(g4.x), MVP_RGB(b4.x), relu_keep_nan(a4.x));
        out[1] = make_float4(MVP_RGB(r4.y), MVP_RGB(g4.y), MVP_RGB(b4.y), relu_keep_nan(a4.y));
        out[2] = make_float4(MVP_RGB(r4.z), MVP_RGB(g4.z), MVP_RGB(b4.z), relu_keep_nan(a4.z));
        out[3] = make_float4(MVP_RGB(r4.w), MVP_RGB(g4.w), MVP_RGB(b4.w), relu_keep_nan(a4.w));
#undef MVP_RGB
    }
}

// grad_tex = 25 * g_rgb * [tplate_rgb > 0], grad_opacity = g_a * [tplate_a > 0]  (relu'(u) = [relu(u) > 0], as torch)
__global__ __launch_bounds__(256) void assemble_bwd_kernel(int N, int nh, int B, const float *__restrict__ tplate,
                                                           const float *__restrict__ gtpl,
                                                           float *__restrict__ gtex, float *__restrict__ gopac) {
    // grid = (x chunks of one image row, image rows, N * B): no 64-bit index arithmetic (three 64-bit div/mod per
    // 128 bytes moved made the first version of these kernels instruction-bound)
    const int S = nh * B, S4 = S >> 2;
    const int X4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (X4 < S4) {
        const int R = blockIdx.y;
        const int n = (int)blockIdx.z / B, z = (int)blockIdx.z - n * B;
        const int X = X4 << 2;
        const size_t plane = (size_t)S * S, rowoff = (size_t)R * S + X;
        const int hy = R / B, y = R - hy * B, wx = X / B, x = X - wx * B;
        const size_t vo = ((((size_t)n * nh * nh + (size_t)hy * nh + wx) * B + z) * B + y) * B + x;
        const float4 *o = reinterpret_cast<const float4 *>(tplate) + vo;
        const float4 *g = reinterpret_cast<const float4 *>(gtpl) + vo;
        float rr[4], gg[4], bb[4], aa[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 ov = o[j], gv = g[j];
            rr[j] = ov.x > 0.f ? gv.x * 25.0f : 0.f;
            gg[j] = ov.y > 0.f ? gv.y * 25.0f : 0.f;
            bb[j] = ov.z > 0.f ? gv.z * 25.0f : 0.f;
            aa[j] = ov.w > 0.f ? gv.w : 0.f;
        }
        float *tp = gtex + ((size_t)n * 3 * B + (size_t)z * 3) * plane + rowoff;
        *reinterpret_cast<float4 *>(tp) = make_float4(rr[0], rr[1], rr[2], rr[3]);
        *reinterpret_cast<float4 *>(tp + plane) = make_float4(gg[0], gg[1], gg[2], gg[3]);
        *reinterpret_cast<float4 *>(tp + 2 * plane) = make_float4(bb[0], bb[1], bb[2], bb[3]);
        *reinterpret_cast<float4 *>(gopac + ((size_t)n * B + z) * plane + rowoff) = make_float4(aa[0], aa[1], aa[2], aa[3]);
    }
}

// ---- frame-broadcast form: ONE decoder output shared by F frames, each frame scaled by its own gain ---------------------
//   tplate[f] = gain[f] * assemble(tex[0], opacity[0])          (all four channels)
// The stand-in decoder of the train leg (trainloop.SlabDecoderStandIn) is the user: it replaces its assemble launch + a
// broadcast multiply (read 1 + F slab tensors, write 1 + F) by one pass (read 1, write F), and in the backward two
// matrix-vector products over the incoming gradient + the assemble backward (read 2 F + 3, write 3) by one pass that reads
// the gradient ONCE (read F + 1, write 1):
//   grad_base = sum_f gain[f] * g[f];  grad_tex = 25 * grad_base * [base_rgb > 0], grad_opacity = grad_base_a * [base_a > 0]
//   grad_gain[f] = sum_voxels <g[f], base>   -> per-workgroup partial sums [blocks, F], summed by the caller (deterministic)
// Same thread mapping as above; the frame loop runs inside the thread with the base slab row in registers.
constexpr int kFramesMax = 1024;  // LDS: 4 waves x F partial sums

__global__ __launch_bounds__(256) void assemble_frames_fwd_kernel(int F, int nh, int B, const float *__restrict__ tex,
                                                                  const float *__restrict__ opac,
                                                                  const float *__restrict__ gain,
                                                                  float *__restrict__ tplate) {
    const int S = nh * B, S4 = S >> 2;
    const int X4 = blockIdx.x * blockDim.x + threadIdx.x;
    if (X4 < S4) {
        const int R = blockIdx.y, z = (int)blockIdx.z;
        const int X = X4 << 2;
        const size_t plane = (size_t)S * S, rowoff = (size_t)R * S + X;
        const float *tp = tex + (size_t)z * 3 * plane + rowoff;
        const float4 r4 = *reinterpret_cast<const float4 *>(tp);
        const float4 g4 = *reinterpret_cast<const float4 *>(tp + plane);
        const float4 b4 = *reinterpret_cast<const float4 *>(tp + 2 * plane);
        const float4 a4 = *reinterpret_cast<const float4 *>(opac + (size_t)z * plane + rowoff);
        const int hy = R / B, y = R - hy * B, wx = X / B, x = X - wx * B;
        const size_t vo = ((((size_t)hy * nh + wx) * B + z) * B + y) * B + x, fstride = (size_t)nh * nh * B * B * B;
#define MVP_RGB(V_) relu_keep_nan(rgb_denorm(V_))
        const float4 v0 = make_float4(MVP_RGB(r4.x), MVP_RGB(g4.x), MVP_RGB(b4.x), relu_keep_nan(a4.x));
        const float4 v1 = make_float4(MVP_RGB(r4.y), MVP_RGB(g4.y), MVP_RGB(b4.y), relu_keep_nan(a4.y));
        const float4 v2 = make_float4(MVP_RGB(r4.z), MVP_RGB(g4.z), MVP_RGB(b4.z), relu_keep_nan(a4.z));
        const float4 v3 = make_float4(MVP_RGB(r4.w), MVP_RGB(g4.w), MVP_RGB(b4.w), relu_keep_nan(a4.w));
#undef MVP_RGB
        float4 *out = reinterpret_cast<float4 *>(tplate) + vo;
        for (int f = 0; f < F; ++f, out += fstride) {
            const float s = gain[f];
            out[0] = make_float4(s * v0.x, s * v0.y, s * v0.z, s * v0.w);
            out[1] = make_float4(s * v1.x, s * v1.y, s * v1.z, s * v1.w);
            out[2] = make_float4(s * v2.x, s * v2.y, s * v2.z, s * v2.w);
            out[3] = make_float4(s * v3.x, s * v3.y, s * v3.z, s * v3.w);
        }
    }
}

__global__ __launch_bounds__(256) void assemble_frames_bwd_kernel(int F, int nh, int B, const float *__restrict__ tex,
                                                                  const float *__restrict__ opac,
                                                                  const float *__restrict__ gain,
                                                                  const float *__restrict__ gtpl,
                                                                  float *__restrict__ gtex, float *__restrict__ gopac,
                                                                  float *__restrict__ gain_partials) {
    extern __shared__ float s_part[];  // [waves][F]
    const int S = nh * B, S4 = S >> 2;
    const int X4 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = X4 < S4;
    const int R = blockIdx.y, z = (int)blockIdx.z;
    const int X = live ? X4 << 2 : 0;
    const size_t plane = (size_t)S * S, rowoff = (size_t)R * S + X;
    const float *tp = tex + (size_t)z * 3 * plane + rowoff;
    const float4 r4 = *reinterpret_cast<const float4 *>(tp);
    const float4 g4 = *reinterpret_cast<const float4 *>(tp + plane);
    const float4 b4 = *reinterpret_cast<const float4 *>(tp + 2 * plane);
    const float4 a4 = *reinterpret_cast<const float4 *>(opac + (size_t)z * plane + rowoff);
    const int hy = R / B, y = R - hy * B, wx = X / B, x = X - wx * B;
    const size_t vo = ((((size_t)hy * nh + wx) * B + z) * B + y) * B + x, fstride = (size_t)nh * nh * B * B * B;
#define MVP_RGB(V_) relu_keep_nan(rgb_denorm(V_))
    float4 v[4] = {make_float4(MVP_RGB(r4.x), MVP_RGB(g4.x), MVP_RGB(b4.x), relu_keep_nan(a4.x)),
                   make_float4(MVP_RGB(r4.y), MVP_RGB(g4.y), MVP_RGB(b4.y), relu_keep_nan(a4.y)),
                   make_float4(MVP_RGB(r4.z), MVP_RGB(g4.z), MVP_RGB(b4.z), relu_keep_nan(a4.z)),
                   make_float4(MVP_RGB(r4.w), MVP_RGB(g4.w), MVP_RGB(b4.w), relu_keep_nan(a4.w))};
#undef MVP_RGB
    float4 acc[4] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f),
                     make_float4(0.f, 0.f, 0.f, 0.f)};
    const float4 *g = reinterpret_cast<const float4 *>(gtpl) + vo;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int f = 0; f < F; ++f, g += fstride) {
        const float s = gain[f];
        float dot = 0.f;
        if (live) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 gv = g[j];
                acc[j].x = fmaf(s, gv.x, acc[j].x), acc[j].y = fmaf(s, gv.y, acc[j].y);
                acc[j].z = fmaf(s, gv.z, acc[j].z), acc[j].w = fmaf(s, gv.w, acc[j].w);
                dot = fmaf(gv.x, v[j].x, fmaf(gv.y, v[j].y, fmaf(gv.z, v[j].z, fmaf(gv.w, v[j].w, dot))));
            }
        }
        dot = wave_sum(dot);
        if (lane == 0) s_part[wave * F + f] = dot;
    }
    if (live) {
        float rr[4], gg[4], bb[4], aa[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rr[j] = v[j].x > 0.f ? acc[j].x * 25.0f : 0.f;
            gg[j] = v[j].y > 0.f ? acc[j].y * 25.0f : 0.f;
            bb[j] = v[j].z > 0.f ? acc[j].z * 25.0f : 0.f;
            aa[j] = v[j].w > 0.f ? acc[j].w : 0.f;
        }
        float *op = gtex + (size_t)z * 3 * plane + rowoff;
        *reinterpret_cast<float4 *>(op) = make_float4(rr[0], rr[1], rr[2], rr[3]);
        *reinterpret_cast<float4 *>(op + plane) = make_float4(gg[0], gg[1], gg[2], gg[3]);
        *reinterpret_cast<float4 *>(op + 2 * plane) = make_float4(bb[0], bb[1], bb[2], bb[3]);
        *reinterpret_cast<float4 *>(gopac + (size_t)z * plane + rowoff) = make_float4(aa[0], aa[1], aa[2], aa[3]);
    }
    __syncthreads();
    const int nwaves = (blockDim.x + 63) >> 6;
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        float t = 0.f;
        for (int w = 0; w < nwaves; ++w) t += s_part[w * F + f];
        gain_partials[blk * F + f] = t;
    }
}

static int assemble_args_ok(int N, int nh, int B) {
    if (N < 0 || nh < 0 || B < 0) return MVP_ERR_BADARG;
    if (B % 4 != 0 && (long long)N * nh * B != 0) return MVP_ERR_UNSUPPORTED;  // 16-byte path needs 4 | B
    return MVP_OK;
}

}  // namespace mvp

extern "C" int mvp_template_assemble_forward(int N, int nh, int B, const float *tex, const float *opacity,
                                             float *tplate, void *stream) {
    using namespace mvp;
    int rc = assemble_args_ok(N, nh, B);
    if (rc != MVP_OK) return rc;
    const long long S = (long long)nh * B, total = (long long)N * B * S * (S / 4);
    if (total == 0) return MVP_OK;
    if (!tex || !opacity || !tplate || !aligned16(tex) || !aligned16(opacity) || !aligned16(tplate)) return MVP_ERR_BADARG;
    if ((long long)N * B > 65535 || S > 65535) return MVP_ERR_UNSUPPORTED;
    const int bx = S / 4 >= 256 ? 256 : (int)(S / 4);
    const dim3 grid((unsigned)((S / 4 + bx - 1) / bx), (unsigned)S, (unsigned)(N * B));
    hipLaunchKernelGGL(assemble_fwd_kernel, grid, dim3(bx), 0, (hipStream_t)stream, N, nh, B, tex, opacity, tplate);
    return launch_status();
}

extern "C" int mvp_template_assemble_backward(int N, int nh, int B, const float *tplate, const float *grad_tplate,
                                              float *grad_tex, float *grad_opacity, void *stream) {
    using namespace mvp;
    int rc = assemble_args_ok(N, nh, B);
    if (rc != MVP_OK) return rc;
    const long long S = (long long)nh * B, total = (long long)N * B * S * (S / 4);
    if (total == 0) return MVP_OK;
    if (!tplate || !grad_tplate || !grad_tex || !grad_opacity || !aligned16(tplate) || !aligned16(grad_tplate) ||
        !aligned16(grad_tex) || !aligned16(grad_opacity))
        return MVP_ERR_BADARG;
    if ((long long)N * B > 65535 || S > 65535) return MVP_ERR_UNSUPPORTED;
    const int bx = S / 4 >= 256 ? 256 : (int)(S / 4);
    const dim3 grid((unsigned)((S / 4 + bx - 1) / bx), (unsigned)S, (unsigned)(N * B));
    hipLaunchKernelGGL(assemble_bwd_kernel, grid, dim3(bx), 0, (hipStream_t)stream, N, nh, B, tplate, grad_tplate,
                       grad_tex, grad_opacity);
    return launch_status();
}

// whole waves only: the per-frame dot products are reduced across the wave, lanes past the row contribute zeros
static int frames_bwd_block(long long S) { return S / 4 >= 256 ? 256 : (int)(((S / 4) + 63) / 64 * 64); }

extern "C" long long mvp_template_assemble_frames_blocks(int nh, int B) {
    const long long S = (long long)nh * B;
    if (S <= 0 || B <= 0 || B % 4 != 0) return 0;
    const long long bx = frames_bwd_block(S);
    return ((S / 4 + bx - 1) / bx) * S * B;
}

extern "C" int mvp_template_assemble_frames_forward(int F, int nh, int B, const float *tex, const float *opacity,
                                                    const float *gain, float *tplate, void *stream) {
    using namespace mvp;
    int rc = assemble_args_ok(F, nh, B);
    if (rc != MVP_OK) return rc;
    const long long S = (long long)nh * B;
    if ((long long)F * B * S == 0) return MVP_OK;
    if (!tex || !opacity || !gain || !tplate || !aligned16(tex) || !aligned16(opacity) || !aligned16(tplate)) return MVP_ERR_BADARG;
    if (B > 65535 || S > 65535) return MVP_ERR_UNSUPPORTED;
    const int bx = S / 4 >= 256 ? 256 : (int)(S / 4);
    const dim3 grid((unsigned)((S / 4 + bx - 1) / bx), (unsigned)S, (unsigned)B);
    hipLaunchKernelGGL(assemble_frames_fwd_kernel, grid, dim3(bx), 0, (hipStream_t)stream, F, nh, B, tex, opacity, gain, tplate);
    return launch_status();
}

extern "C" int mvp_template_assemble_frames_backward(int F, int nh, int B, const float *tex, const float *opacity,
                                                     const float *gain, const float *grad_tplate, float *grad_tex,
                                                     float *grad_opacity, float *gain_partials, void *stream) {
    using namespace mvp;
    int rc = assemble_args_ok(F, nh, B);
    if (rc != MVP_OK) return rc;
    const long long S = (long long)nh * B;
    if ((long long)B * S == 0) return MVP_OK;
    if (F > kFramesMax) return MVP_ERR_UNSUPPORTED;
    if (!tex || !opacity || !grad_tex || !grad_opacity || !aligned16(tex) || !aligned16(opacity) || !aligned16(grad_tex) ||
        !aligned16(grad_opacity))
        return MVP_ERR_BADARG;
    if (F > 0 && (!gain || !grad_tplate || !gain_partials || !aligned16(grad_tplate))) return MVP_ERR_BADARG;
    if (B > 65535 || S > 65535) return MVP_ERR_UNSUPPORTED;
    const int bx = frames_bwd_block(S);
    const dim3 grid((unsigned)((S / 4 + bx - 1) / bx), (unsigned)S, (unsigned)B);
    const size_t lds = (size_t)((bx + 63) / 64) * (size_t)(F > 0 ? F : 1) * sizeof(float);
    hipLaunchKernelGGL(assemble_frames_bwd_kernel, grid, dim3(bx), lds, (hipStream_t)stream, F, nh, B, tex, opacity, gain,
                       grad_tplate, grad_tex, grad_opacity, gain_partials);
    return launch_status();
}
