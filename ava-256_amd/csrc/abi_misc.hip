// abi_misc.hip -- version / error-string / device-query entry points of the C ABI.
#include <string.h>

#include "mvp_host.h"

extern "C" int mvp_abi_version(void) { return MVP_ABI_VERSION; }

extern "C" const char *mvp_error_string(int code) {
    switch (code) {
        case MVP_OK: return "ok";
        case MVP_ERR_BADARG: return "bad argument (null / misaligned pointer, bad size or non-finite scalar)";
        case MVP_ERR_UNSUPPORTED: return "shape not supported by this build";
        case MVP_ERR_NODEVICE: return "no usable HIP device";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}

extern "C" int mvp_device_arch(int device, char *buf, int buflen) {
    if (!buf || buflen <= 0) return MVP_ERR_BADARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
        (void)hipGetLastError();
        return MVP_ERR_NODEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MVP_ERR_NODEVICE;
    strncpy(buf, prop.gcnArchName, (size_t)buflen - 1);
    buf[buflen - 1] = 0;
    return MVP_OK;
}

// ---- NHWC -> NCHW split of the march result (row A14 / N1, second half) ------------------------------------------
// The reference's Raymarcher permutes the [N,H,W,4] march output to NCHW and makes two contiguous copies,
// rayrgb = rgba[:, :3] and rayalpha = rgba[:, 3:4] (models/raymarchers/mvpraymarcher.py:50-51); autograd then runs the
// slice / copy backward as several more passes.  One pass each way here: a thread owns one pixel, reads (writes) its
// 16-byte RGBA and writes (reads) four plane elements -- both sides fully coalesced.  Pure data movement: bit-exact.
namespace mvp {
__global__ __launch_bounds__(256) void rgba_split_fwd_kernel(const float4 *__restrict__ rgba, size_t hw, size_t total,
                                                             float *__restrict__ rgb, float *__restrict__ alpha) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    size_t n, p;
    if (total < 0x7fffffffull) {  // (uniform) 32-bit division instead of ~80 instructions of 64-bit division
        const unsigned n32 = (unsigned)g / (unsigned)hw;
        n = n32, p = (unsigned)g - n32 * (unsigned)hw;
    } else {
        n = g / hw, p = g - n * hw;
    }
    const float4 v = rgba[g];
    float *o = rgb + n * 3 * hw + p;
    o[0] = v.x, o[hw] = v.y, o[2 * hw] = v.z;
    alpha[g] = v.w;
}
__global__ __launch_bounds__(256) void rgba_split_bwd_kernel(const float *__restrict__ g_rgb, const float *__restrict__ g_alpha,
                                                             size_t hw, size_t total, float4 *__restrict__ g_rgba) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    size_t n, p;
    if (total < 0x7fffffffull) {
        const unsigned n32 = (unsigned)g / (unsigned)hw;
        n = n32, p = (unsigned)g - n32 * (unsigned)hw;
    } else {
        n = g / hw, p = g - n * hw;
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g_rgb) {
        const float *i = g_rgb + n * 3 * hw + p;
        v.x = i[0], v.y = i[hw], v.z = i[2 * hw];
    }
    if (g_alpha) v.w = g_alpha[g];
    g_rgba[g] = v;
}
}  // namespace mvp

extern "C" int mvp_rgba_split_forward(int N, int H, int W, const float *rayrgba, float *rayrgb, float *rayalpha,
                                      void *stream) {
    if (N < 0 || H < 0 || W < 0) return MVP_ERR_BADARG;
    const size_t hw = (size_t)H * W, total = hw * (size_t)N;
    if (total == 0) return MVP_OK;
    if (!rayrgba || !rayrgb || !rayalpha || !mvp::aligned16(rayrgba)) return MVP_ERR_BADARG;
    if ((total + 255) / 256 > 0x7fffffffull) return MVP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(mvp::rgba_split_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(rayrgba), hw, total, rayrgb, rayalpha);
    return mvp::launch_status();
}

extern "C" int mvp_rgba_split_backward(int N, int H, int W, const float *grad_rayrgb, const float *grad_rayalpha,
                                       float *grad_rayrgba, void *stream) {
    if (N < 0 || H < 0 || W < 0) return MVP_ERR_BADARG;
    const size_t hw = (size_t)H * W, total = hw * (size_t)N;
    if (total == 0) return MVP_OK;
    if (!grad_rayrgba || !mvp::aligned16(grad_rayrgba)) return MVP_ERR_BADARG;
    if ((total + 255) / 256 > 0x7fffffffull) return MVP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(mvp::rgba_split_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       grad_rayrgb, grad_rayalpha, hw, total, reinterpret_cast<float4 *>(grad_rayrgba));
    return mvp::launch_status();
}

// ---- demand statistics of the forward -> backward packet lists -----------------------------------------------------
// The forward's per-primitive counters keep counting past the list capacity; the operator sizes the next call's lists
// from them (ava-256_amd/mvpraymarch.py: note_list_demand).  One pass: a 256-bin histogram of min(count, 2047) / 8 and the
// maximum, so that the host can tell an outlier (one image-filling primitive) from a shift of the whole distribution.
namespace mvp {
__global__ __launch_bounds__(256) void list_demand_kernel(const uint32_t *__restrict__ counts, long long n,
                                                          uint32_t *__restrict__ hist) {
    __shared__ uint32_t s_h[257];
    for (int i = threadIdx.x; i < 257; i += 256) s_h[i] = 0u;
    __syncthreads();
    uint32_t mx = 0u;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const uint32_t c = counts[i] & 0x3fffffffu;  // (bits 30-31 are marks of a backward)
        mx = max(mx, c);
        atomicAdd(&s_h[min(c, 2047u) >> 3], 1u);
    }
    atomicMax(&s_h[256], mx);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 256)
        if (s_h[i]) atomicAdd(hist + i, s_h[i]);
    if (threadIdx.x == 0 && s_h[256]) atomicMax(hist + 256, s_h[256]);
}
}  // namespace mvp

extern "C" int mvp_list_demand(const uint32_t *primlist_count, long long nprims, uint32_t *hist, void *stream) {
    if (nprims < 0 || !hist) return MVP_ERR_BADARG;
    hipError_t e = hipMemsetAsync(hist, 0, 257 * sizeof(uint32_t), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    if (nprims == 0) return MVP_OK;
    if (!primlist_count) return MVP_ERR_BADARG;
    long long blocks = (nprims + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(mvp::list_demand_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, primlist_count,
                       nprims, hist);
    return mvp::launch_status();
}
