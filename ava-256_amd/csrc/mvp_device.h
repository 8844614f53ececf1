// mvp_device.h -- small device-side helpers shared by the gfx950 kernels (wave64, CDNA4 only).
// Written for this project; the reference's GLSL-style helper header
// (/root/reference/extensions/include/helper_math.h) is not used.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mvp {

constexpr int kWave = 64;  // CDNA wavefront width; every kernel here is written for exactly this

struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 ld3(const float *p) { return f3{p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float *p, f3 v) {
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}
__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// base + 32-bit BYTE offset: lets the backend address with a scalar base and a zero-extended vector offset
// (global_load ... v_off, s[base:base+1]) instead of carrying a 64-bit address in two VGPRs per stream
template <class T>
__device__ __forceinline__ const T *at_bytes(const void *base, uint32_t byte_off) {
    return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}

// ---- pixel -> ray (utils_kernel.cu:12-52) -----------------------------------------------------------------------
// One statement of the ray arithmetic for every kernel that makes rays (raydirs_kernel, and the march when it is handed
// cameras instead of ray tensors): explicit fused operations under `fp contract(off)`, IEEE division and square root
// (hipcc's default), so the same camera and pixel give the same bits wherever this is inlined.
struct CamRay {
    f3 o, d;
    float tmin, tmax;
};
__device__ __forceinline__ CamRay ray_from_camera(f3 campos, const float *__restrict__ R /*[3,3] rows*/, float fx,
                                                  float fy, float cx, float cy, float px, float py, float volradius) {
#pragma clang fp contract(off)
    // (round 6) Reciprocals are v_rcp_f32 / v_rsq_f32 (1 ulp) and every quotient a product with one: the IEEE division
    // sequence is ~10 VALU instructions and this function had twelve of them -- raydirs_kernel, a streaming-store kernel, was
    // bound by its VALU at 0.20 ms (3.3 TB/s).  The reference's own build divides approximately too (-use_fast_math,
    // extensions/utils/setup.py); the ray tests hold 1e-6 / 2e-6 / 2e-5 on origin / direction / interval.  The march that makes
    // its rays itself (mvp_march_forward_cams) runs this same function, so the two stay bit-identical.
    CamRay c;
    const float ivr = __builtin_amdgcn_rcpf(volradius);
    c.o = mk3(campos.x * ivr, campos.y * ivr, campos.z * ivr);  // utils_kernel.cu:32
    const float qx = (px - cx) * __builtin_amdgcn_rcpf(fx), qy = (py - cy) * __builtin_amdgcn_rcpf(fy);
    f3 d = mk3(__builtin_fmaf(R[3], qy, R[0] * qx) + R[6], __builtin_fmaf(R[4], qy, R[1] * qx) + R[7],
               __builtin_fmaf(R[5], qy, R[2] * qx) + R[8]);
    const float inv = __builtin_amdgcn_rsqf(__builtin_fmaf(d.z, d.z, __builtin_fmaf(d.y, d.y, d.x * d.x)));
    d = mk3(d.x * inv, d.y * inv, d.z * inv);
    c.d = d;
    const f3 id = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    const f3 t1 = mk3((-1.f - c.o.x) * id.x, (-1.f - c.o.y) * id.y, (-1.f - c.o.z) * id.z);
    const f3 t2 = mk3((1.f - c.o.x) * id.x, (1.f - c.o.y) * id.y, (1.f - c.o.z) * id.z);
    c.tmin = fmaxf(max3f(fminf(t1.x, t2.x), fminf(t1.y, t2.y), fminf(t1.z, t2.z)), 0.f);
    c.tmax = min3f(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y), fmaxf(t1.z, t2.z));
    return c;
}

// ---- wave64 cross-lane helpers ------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ unsigned long long lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// Wave64 reductions without LDS traffic: a 4-step DPP butterfly reduces each row of 16 lanes (every lane of the row
// ends up with the row result), then the four row results are read with v_readlane and combined.  A ds_bpermute
// butterfly (what __shfl_xor compiles to) costs 6 dependent LDS round trips per reduction; the exact-test loop of
// the march does two reductions per candidate primitive.
//   DPP controls (gfx9): quad_perm:[1,0,3,2] = 0xB1, quad_perm:[2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140
#define MVP_DPP_I(v, ctrl) __builtin_amdgcn_update_dpp((v), (v), (ctrl), 0xf, 0xf, false)
__device__ __forceinline__ float dpp_f(float v, int which) {
    const int i = __float_as_int(v);
    int r;
    switch (which) {
        case 0: r = MVP_DPP_I(i, 0xB1); break;
        case 1: r = MVP_DPP_I(i, 0x4E); break;
        case 2: r = MVP_DPP_I(i, 0x141); break;
        default: r = MVP_DPP_I(i, 0x140); break;
    }
    return __int_as_float(r);
}
__device__ __forceinline__ int dpp_i(int i, int which) {
    switch (which) {
        case 0: return MVP_DPP_I(i, 0xB1);
        case 1: return MVP_DPP_I(i, 0x4E);
        case 2: return MVP_DPP_I(i, 0x141);
        default: return MVP_DPP_I(i, 0x140);
    }
}
#undef MVP_DPP_I
__device__ __forceinline__ float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// min / max over the wave in 6 fused DPP instructions + 1 v_readlane (idempotent operations only): an inclusive scan inside
// every row of 16 lanes (row_shr 1, 2, 4, 8: lane 15 of a row ends up with the row's result; a lane without a source is
// disabled and keeps its value), then row_bcast:15 (lane 15 of a row -> the next row; rows 1 and 3 take it) and row_bcast:31
// (lane 31 -> rows 2 and 3): lane 63 holds the result.  The butterfly form (wave_sum below) costs 4 x (v_mov_dpp + op) + 4
// v_readlane + 3 operations = 15 VALU instructions -- hipcc does not fold __builtin_amdgcn_update_dpp into the operation --
// and the forward's exact test does two reductions per listed primitive, its packet bounds eight to fourteen per packet.
// Written as one asm block: a DPP read of a VGPR the previous VALU instruction wrote needs two wait states (s_nop 1), which
// the compiler's hazard recogniser does not insert inside inline assembly.  Every caller runs these with all 64 lanes
// enabled (inactive rays pass neutral values).
#define MVP_WAVE_SCAN(OPC_)                                                                    \
    asm("s_nop 1\n\t" OPC_ " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"               \
        "s_nop 1\n\t" OPC_ " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"               \
        "s_nop 1\n\t" OPC_ " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"               \
        "s_nop 1\n\t" OPC_ " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"               \
        "s_nop 1\n\t" OPC_ " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"            \
        "s_nop 1\n\t" OPC_ " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"            \
        "s_nop 1"                                                                              \
        : "+v"(v))
__device__ __forceinline__ float wave_min(float v) {
    MVP_WAVE_SCAN("v_min_f32_dpp");
    return rl_f(v, 63);
}
__device__ __forceinline__ float wave_max(float v) {
    MVP_WAVE_SCAN("v_max_f32_dpp");
    return rl_f(v, 63);
}
__device__ __forceinline__ int wave_min(int v) {
    MVP_WAVE_SCAN("v_min_i32_dpp");
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max(int v) {
    MVP_WAVE_SCAN("v_max_i32_dpp");
    return __builtin_amdgcn_readlane(v, 63);
}
#undef MVP_WAVE_SCAN
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int w = 0; w < 4; ++w) v += dpp_f(v, w);
    return (rl_f(v, 0) + rl_f(v, 16)) + (rl_f(v, 32) + rl_f(v, 48));
}
// Twelve sums over the wave at once (the backward's pose sums): the same row_shr / row_bcast scan as wave_min / wave_max with
// v_add_f32 -- an inclusive prefix sum inside every row of 16 lanes, then the row totals carried over -- but with the twelve
// chains INTERLEAVED, so that every DPP read is twelve instructions behind the write it depends on and needs no wait states:
// 72 VALU instructions for twelve sums, where twelve wave_sum() calls cost 12 x 15 plus the moves around them (round 6: the
// ISA census of bwd_prim_kernel showed 253 VALU instructions per wave in this reduction, 6 % of all the kernel executes).
// Lane 63 ends up with the totals; all 64 lanes must be enabled.
__device__ __forceinline__ void wave_sum12_lane63(float (&v)[12]) {
#define MVP_ADD12(CTRL_)                                                                                  \
    "v_add_f32_dpp %0, %0, %0 " CTRL_ "\n\tv_add_f32_dpp %1, %1, %1 " CTRL_ "\n\t"                        \
    "v_add_f32_dpp %2, %2, %2 " CTRL_ "\n\tv_add_f32_dpp %3, %3, %3 " CTRL_ "\n\t"                        \
    "v_add_f32_dpp %4, %4, %4 " CTRL_ "\n\tv_add_f32_dpp %5, %5, %5 " CTRL_ "\n\t"                        \
    "v_add_f32_dpp %6, %6, %6 " CTRL_ "\n\tv_add_f32_dpp %7, %7, %7 " CTRL_ "\n\t"                        \
    "v_add_f32_dpp %8, %8, %8 " CTRL_ "\n\tv_add_f32_dpp %9, %9, %9 " CTRL_ "\n\t"                        \
    "v_add_f32_dpp %10, %10, %10 " CTRL_ "\n\tv_add_f32_dpp %11, %11, %11 " CTRL_ "\n\t"
    asm("s_nop 1\n\t" MVP_ADD12("row_shr:1 row_mask:0xf bank_mask:0xf") MVP_ADD12("row_shr:2 row_mask:0xf bank_mask:0xf")
        MVP_ADD12("row_shr:4 row_mask:0xf bank_mask:0xf") MVP_ADD12("row_shr:8 row_mask:0xf bank_mask:0xf")
        MVP_ADD12("row_bcast:15 row_mask:0xa bank_mask:0xf") MVP_ADD12("row_bcast:31 row_mask:0xc bank_mask:0xf") "s_nop 1"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
          "+v"(v[9]), "+v"(v[10]), "+v"(v[11]));
#undef MVP_ADD12
}
// make a value the compiler cannot prove uniform live in an SGPR
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// v_exp_f32 / v_log_f32 (about 1 ulp); the reference uses the equivalent CUDA fast intrinsics
// __expf / __powf (primsampler.h:48-51, built with -use_fast_math, extensions/mvpraymarch/setup.py:27)
// Load through the constant address space: with a wave-uniform address this is an s_load (result in SGPRs, no VGPR
// and no vector-memory slot).  Only for data no kernel in flight writes (inputs, the previous kernel's outputs).
template <class T>
__device__ __forceinline__ T cload(const T *p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T *>(reinterpret_cast<uintptr_t>(p));
}

// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division sequence; used where a conservative slack or the
// exact inside test downstream absorbs the last bit (slab-interval tests, step-index ranges)
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_exp(float x) { return fast_exp2(x * 1.44269504088896341f); }
// |x|^e for x in [0,1], e > 0 : 0^e = exp2(e * -inf) = 0
__device__ __forceinline__ float fast_pow(float ax, float e) { return fast_exp2(e * fast_log2(ax)); }

}  // namespace mvp
