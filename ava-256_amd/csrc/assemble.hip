// assemble.hip -- decoder -> raymarch hand-off: builds the channels-last RGBA slab tensor the march reads directly
// from the two conv-decoder outputs, in one pass.
//
// What it replaces in the reference (three separate places, ~4 full passes over a 134 MB/image tensor in eager PyTorch):
//   models/decoders/rgb.py:137-143        rgb.view(N, B, 3, h, B, w, B).permute(0,3,5,1,4,6,2).reshape(N, h*w, B,B,B, 3)
//   models/decoders/geometry.py:183-185   the same with 1 channel for the opacity
//   models/decoders/assembler.py:261      template = cat([relu(rgb * 25 + 100), relu(alpha)], dim=-1)
// i.e.  tplate[n, hy*nh + wx, z, y, x, c] = relu(tex[n, z*3 + c, hy*B + y, wx*B + x] * 25 + 100)   (c < 3)
//       tplate[n, hy*nh + wx, z, y, x, 3] = relu(opacity[n, z, hy*B + y, wx*B + x])
// Pure data movement: HBM-bound, 16 B read + 16 B written per voxel forward, 32 + 16 backward.
//
// Mapping for CDNA4: one thread per texel (see the note above the kernels): 4-byte plane accesses (a wave covers 256
// contiguous bytes per plane) and one 16-byte slab access per thread, eight consecutive lanes = one whole 128-byte slab row.
// No LDS, no atomics (the frame-broadcast backward reduces its per-frame dot products per wave).
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

// Thread mapping of all four kernels: ONE texel (four channels) per thread, lanes consecutive along an image row.  Eight
// consecutive lanes hold one slab row = one whole 128-byte line of the slab tensor per 16-byte access instruction; the plane
// accesses are 4-byte, 256 contiguous bytes per wave and plane.  (Rounds 1-4 gave a thread four consecutive texels -- 16-byte
// plane accesses, but a slab row written as 2 lanes x 4 instructions of 16 bytes each: 3.2 TB/s where this mapping reaches
// 5.9 TB/s on the frame-broadcast forward at C2, profiles/r04_assemble_mapping_ab.txt.)
// grid = (row chunks, image rows, frames * depth): no 64-bit index arithmetic.
struct Texel {
    size_t plane, rowoff, vo;  // plane size, offset inside a plane, float4 index inside one frame's slab tensor
    int z;
};
__device__ __forceinline__ bool texel_of_thread(int nh, int B, int zz, Texel &t) {
    const int S = nh * B;
    const int X = blockIdx.x * blockDim.x + threadIdx.x;
    t.z = zz;
    if (X >= S) {
        t.plane = (size_t)S * S, t.rowoff = 0, t.vo = 0;
        return false;
    }
    const int R = blockIdx.y;
    t.plane = (size_t)S * S, t.rowoff = (size_t)R * S + X;
    const int hy = R / B, y = R - hy * B, wx = X / B, x = X - wx * B;
    t.vo = ((((size_t)hy * nh + wx) * B + zz) * B + y) * B + x;
    return true;
}
__device__ __forceinline__ float4 texel_value(const float *__restrict__ tex_n, const float *__restrict__ opac_n, const Texel &t) {
    const float *tp = tex_n + (size_t)t.z * 3 * t.plane + t.rowoff;
    return make_float4(relu_keep_nan(rgb_denorm(tp[0])), relu_keep_nan(rgb_denorm(tp[t.plane])),
                       relu_keep_nan(rgb_denorm(tp[2 * t.plane])), relu_keep_nan(opac_n[(size_t)t.z * t.plane + t.rowoff]));
}

__global__ __launch_bounds__(256) void assemble_fwd_kernel(int N, int nh, int B, const float *__restrict__ tex,
                                                           const float *__restrict__ opac,
                                                           float *__restrict__ tplate) {
    const int n = (int)blockIdx.z / B;
    Texel t;
    if (!texel_of_thread(nh, B, (int)blockIdx.z - n * B, t)) return;
    const size_t fstride = (size_t)nh * nh * B * B * B;
    reinterpret_cast<float4 *>(tplate)[(size_t)n * fstride + t.vo] =
        texel_value(tex + (size_t)n * 3 * B * t.plane, opac + (size_t)n * B * t.plane, t);
}

// ---- half-precision slabs for the opt-in render path (march_common.h: sample_slab_h) -------------------------------------
// fp16 RGBA, 8 bytes per voxel, round to nearest even (v_cvt_f16_f32); values beyond 65504 become +-inf like any fp16 cast
// (slab values are relu(x * 25 + 100): hundreds).  Same thread mapping as above: eight consecutive lanes write one 64-byte
// slab row.  Written straight from the decoder outputs, the half slab costs 16 B read + 8 B written per voxel; converting
// an existing fp32 slab tensor (mvp_template_to_half) 16 + 8 as well.
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ h4 to_half4(float4 v) { return h4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w}; }

__global__ __launch_bounds__(256) void assemble_fwd_half_kernel(int N, int nh, int B, const float *__restrict__ tex,
                                                                const float *__restrict__ opac, h4 *__restrict__ tplate) {
    const int n = (int)blockIdx.z / B;
    Texel t;
    if (!texel_of_thread(nh, B, (int)blockIdx.z - n * B, t)) return;
    const size_t fstride = (size_t)nh * nh * B * B * B;
    tplate[(size_t)n * fstride + t.vo] = to_half4(texel_value(tex + (size_t)n * 3 * B * t.plane, opac + (size_t)n * B * t.plane, t));
}

// two voxels per thread: 32 bytes read, 16 written
__global__ __launch_bounds__(256) void template_to_half_kernel(size_t pairs, const float4 *__restrict__ src,
                                                               float4 *__restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pairs) return;
    const float4 a = src[2 * i], b = src[2 * i + 1];
    union { h4 h[2]; float4 f; } u;
    u.h[0] = to_half4(a), u.h[1] = to_half4(b);
    dst[i] = u.f;
}

// grad_tex = 25 * g_rgb * [tplate_rgb > 0], grad_opacity = g_a * [tplate_a > 0]  (relu'(u) = [relu(u) > 0], as torch)
__global__ __launch_bounds__(256) void assemble_bwd_kernel(int N, int nh, int B, const float *__restrict__ tplate,
                                                           const float *__restrict__ gtpl,
                                                           float *__restrict__ gtex, float *__restrict__ gopac) {
    const int n = (int)blockIdx.z / B;
    Texel t;
    if (!texel_of_thread(nh, B, (int)blockIdx.z - n * B, t)) return;
    const size_t fstride = (size_t)nh * nh * B * B * B;
    const float4 ov = reinterpret_cast<const float4 *>(tplate)[(size_t)n * fstride + t.vo];
    const float4 gv = reinterpret_cast<const float4 *>(gtpl)[(size_t)n * fstride + t.vo];
    float *tp = gtex + ((size_t)n * 3 * B + (size_t)t.z * 3) * t.plane + t.rowoff;
    tp[0] = ov.x > 0.f ? gv.x * 25.0f : 0.f;
    tp[t.plane] = ov.y > 0.f ? gv.y * 25.0f : 0.f;
    tp[2 * t.plane] = ov.z > 0.f ? gv.z * 25.0f : 0.f;
    gopac[((size_t)n * B + t.z) * t.plane + t.rowoff] = ov.w > 0.f ? gv.w : 0.f;
}

// ---- frame-broadcast form: ONE decoder output shared by F frames, each frame scaled by its own gain ---------------------
//   tplate[f] = gain[f] * assemble(tex[0], opacity[0])          (all four channels)
// The stand-in decoder of the train leg (trainloop.SlabDecoderStandIn) is the user: it replaces its assemble launch + a
// broadcast multiply (read 1 + F slab tensors, write 1 + F) by one pass (read 1, write F), and in the backward two
// matrix-vector products over the incoming gradient + the assemble backward (read 2 F + 3, write 3) by one pass that reads
// the gradient ONCE (read F + 1, write 1):
//   grad_base = sum_f gain[f] * g[f];  grad_tex = 25 * grad_base * [base_rgb > 0], grad_opacity = grad_base_a * [base_a > 0]
//   grad_gain[f] = sum_voxels <g[f], base>   -> per-workgroup partial sums [blocks, F], summed by the caller (deterministic)
// The frame loop runs inside the thread with the base texel in registers.
constexpr int kFramesMax = 1024;  // LDS: 4 waves x F partial sums

__global__ __launch_bounds__(256) void assemble_frames_fwd_kernel(int F, int nh, int B, const float *__restrict__ tex,
                                                                  const float *__restrict__ opac,
                                                                  const float *__restrict__ gain,
                                                                  float *__restrict__ tplate) {
    Texel t;
    if (!texel_of_thread(nh, B, (int)blockIdx.z, t)) return;
    const float4 v = texel_value(tex, opac, t);
    const size_t fstride = (size_t)nh * nh * B * B * B;
    float4 *out = reinterpret_cast<float4 *>(tplate) + t.vo;
    for (int f = 0; f < F; ++f, out += fstride) {
        const float s = gain[f];
        *out = make_float4(s * v.x, s * v.y, s * v.z, s * v.w);
    }
}

__global__ __launch_bounds__(256) void assemble_frames_bwd_kernel(int F, int nh, int B, const float *__restrict__ tex,
                                                                  const float *__restrict__ opac,
                                                                  const float *__restrict__ gain,
                                                                  const float *__restrict__ gtpl,
                                                                  float *__restrict__ gtex, float *__restrict__ gopac,
                                                                  float *__restrict__ gain_partials) {
    extern __shared__ float s_part[];  // [waves][F]
    Texel t;
    const bool live = texel_of_thread(nh, B, (int)blockIdx.z, t);
    const float4 v = texel_value(tex, opac, t);   // (a dead thread reads texel 0 of the row: a valid address)
    const size_t fstride = (size_t)nh * nh * B * B * B;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 *g = reinterpret_cast<const float4 *>(gtpl) + t.vo;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int f = 0; f < F; ++f, g += fstride) {
        const float s = gain[f];
        float dot = 0.f;
        if (live) {
            const float4 gv = *g;
            acc.x = fmaf(s, gv.x, acc.x), acc.y = fmaf(s, gv.y, acc.y), acc.z = fmaf(s, gv.z, acc.z), acc.w = fmaf(s, gv.w, acc.w);
            dot = fmaf(gv.x, v.x, fmaf(gv.y, v.y, fmaf(gv.z, v.z, gv.w * v.w)));
        }
        dot = wave_sum(dot);
        if (lane == 0) s_part[wave * F + f] = dot;
    }
    if (live) {
        float *op = gtex + (size_t)t.z * 3 * t.plane + t.rowoff;
        op[0] = v.x > 0.f ? acc.x * 25.0f : 0.f;
        op[t.plane] = v.y > 0.f ? acc.y * 25.0f : 0.f;
        op[2 * t.plane] = v.z > 0.f ? acc.z * 25.0f : 0.f;
        gopac[(size_t)t.z * t.plane + t.rowoff] = v.w > 0.f ? acc.w : 0.f;
    }
    __syncthreads();
    const int nwaves = (blockDim.x + 63) >> 6;
    const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        float tsum = 0.f;
        for (int w = 0; w < nwaves; ++w) tsum += s_part[w * F + f];
        gain_partials[blk * F + f] = tsum;
    }
}

static int assemble_args_ok(int N, int nh, int B) {
    if (N < 0 || nh < 0 || B < 0) return MVP_ERR_BADARG;
    return MVP_OK;
}
// whole waves (the frame-broadcast backward reduces across the wave; lanes past the row contribute zeros)
static int assemble_block(long long S) { return S >= 256 ? 256 : (int)((S + 63) / 64 * 64); }

}  // namespace mvp

extern "C" int mvp_template_assemble_forward(int N, int nh, int B, const float *tex, const float *opacity,
                                             float *tplate, void *stream) {
    using namespace mvp;
    int rc = assemble_args_ok(N, nh, B);
    if (rc != MVP_OK) return rc;
    const long long S = (long long)nh * B;
    if ((long long)N * B * S == 0) return MVP_OK;
    if (!tex || !opacity || !tplate || !aligned16(tplate)) return MVP_ERR_BADARG;
    if ((long long)N * B > 65535 || S > 65535) return MVP_ERR_UNSUPPORTED;
    const int bx = assemble_block(S);
    const dim3 grid((unsigned)((S + bx - 1) / bx), (unsigned)S, (unsigned)(N * B));
    hipLaunchKernelGGL(assemble_fwd_kernel, grid, dim3(bx), 0, (hipStream_t)stream, N, nh, B, tex, opacity, tplate);
    return launch_status();
}

extern "C" int mvp_template_assemble_backward(int N, int nh, int B, const float *tplate, const float *grad_tplate,
                                              float *grad_tex, float *grad_opacity, void *stream) {
    using namespace mvp;
    int rc = assemble_args_ok(N, nh, B);
    if (rc != MVP_OK) return rc;
    const long long S = (long long)nh * B;
    if ((long long)N * B * S == 0) return MVP_OK;
    if (!tplate || !grad_tplate || !grad_tex || !grad_opacity || !aligned16(tplate) || !aligned16(grad_tplate))
        return MVP_ERR_BADARG;
    if ((long long)N * B > 65535 || S > 65535) return MVP_ERR_UNSUPPORTED;
    const int bx = assemble_block(S);
    const dim3 grid((unsigned)((S + bx - 1) / bx), (unsigned)S, (unsigned)(N * B));
    hipLaunchKernelGGL(assemble_bwd_kernel, grid, dim3(bx), 0, (hipStream_t)stream, N, nh, B, tplate, grad_tplate,
                       grad_tex, grad_opacity);
    return launch_status();
}

extern "C" long long mvp_template_assemble_frames_blocks(int nh, int B) {
    const long long S = (long long)nh * B;
    if (S <= 0 || B <= 0) return 0;
    const long long bx = mvp::assemble_block(S);
    return ((S + bx - 1) / bx) * S * B;
}

extern "C" int mvp_template_assemble_frames_forward(int F, int nh, int B, const float *tex, const float *opacity,
                                                    const float *gain, float *tplate, void *stream) {
    using namespace mvp;
    int rc = assemble_args_ok(F, nh, B);
    if (rc != MVP_OK) return rc;
    const long long S = (long long)nh * B;
    if ((long long)F * B * S == 0) return MVP_OK;
    if (!tex || !opacity || !gain || !tplate || !aligned16(tplate)) return MVP_ERR_BADARG;
    if (B > 65535 || S > 65535) return MVP_ERR_UNSUPPORTED;
    const int bx = assemble_block(S);
    const dim3 grid((unsigned)((S + bx - 1) / bx), (unsigned)S, (unsigned)B);
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
    if (!tex || !opacity || !grad_tex || !grad_opacity) return MVP_ERR_BADARG;
    if (F > 0 && (!gain || !grad_tplate || !gain_partials || !aligned16(grad_tplate))) return MVP_ERR_BADARG;
    if (B > 65535 || S > 65535) return MVP_ERR_UNSUPPORTED;
    const int bx = assemble_block(S);
    const dim3 grid((unsigned)((S + bx - 1) / bx), (unsigned)S, (unsigned)B);
    const size_t lds = (size_t)(bx / 64) * (size_t)(F > 0 ? F : 1) * sizeof(float);
    hipLaunchKernelGGL(assemble_frames_bwd_kernel, grid, dim3(bx), lds, (hipStream_t)stream, F, nh, B, tex, opacity, gain,
                       grad_tplate, grad_tex, grad_opacity, gain_partials);
    return launch_status();
}

extern "C" int mvp_template_assemble_forward_half(int N, int nh, int B, const float *tex, const float *opacity,
                                                  void *tplate_half, void *stream) {
    using namespace mvp;
    int rc = assemble_args_ok(N, nh, B);
    if (rc != MVP_OK) return rc;
    const long long S = (long long)nh * B;
    if ((long long)N * B * S == 0) return MVP_OK;
    if (!tex || !opacity || !tplate_half || !aligned16(tplate_half)) return MVP_ERR_BADARG;
    if ((long long)N * B > 65535 || S > 65535) return MVP_ERR_UNSUPPORTED;
    const int bx = assemble_block(S);
    const dim3 grid((unsigned)((S + bx - 1) / bx), (unsigned)S, (unsigned)(N * B));
    hipLaunchKernelGGL(assemble_fwd_half_kernel, grid, dim3(bx), 0, (hipStream_t)stream, N, nh, B, tex, opacity,
                       reinterpret_cast<h4 *>(tplate_half));
    return launch_status();
}

extern "C" int mvp_template_to_half(long long voxels, const float *tplate, void *tplate_half, void *stream) {
    using namespace mvp;
    if (voxels < 0 || (voxels & 1)) return MVP_ERR_BADARG;  // (slabs have an even number of voxels)
    if (voxels == 0) return MVP_OK;
    if (!tplate || !tplate_half || !aligned16(tplate) || !aligned16(tplate_half)) return MVP_ERR_BADARG;
    const size_t pairs = (size_t)voxels / 2;
    if ((pairs + 255) / 256 > 0x7fffffffull) return MVP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(template_to_half_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pairs,
                       reinterpret_cast<const float4 *>(tplate), reinterpret_cast<float4 *>(tplate_half));
    return launch_status();
}
