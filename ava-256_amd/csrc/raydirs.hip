// raydirs.hip -- pixel -> ray generation for gfx950.
// Semantics: /root/reference/extensions/utils/utils_kernel.cu:12-52 (forward; there is no backward,
// utils_kernel.cu:54-95 writes nothing).  Pure streaming-store kernel: 32 B written per ray, <= 8 B read.
//
// Layout choice for CDNA4: one wave owns 64 CONSECUTIVE rays of the flattened [N*H*W] index space, so its
// three output streams are contiguous runs of 768 B / 768 B / 512 B.  The float3 streams are staged through
// LDS and written back as 16-byte dwordx4 stores (three per 4 rays) instead of 12-byte strided stores.
#include "mvp_device.h"
#include "mvp_host.h"

namespace mvp {

constexpr int kRdBlock = 256;

__global__ __launch_bounds__(kRdBlock) void raydirs_kernel(int N, int H, int W, const float *__restrict__ campos,
                                                           const float *__restrict__ camrot,
                                                           const float *__restrict__ focal,
                                                           const float *__restrict__ princpt,
                                                           const float *__restrict__ pixelcoords, float volradius,
                                                           float *__restrict__ raypos, float *__restrict__ raydir,
                                                           float *__restrict__ tminmax, int vec_ok) {
    __shared__ float s_pos[kRdBlock * 3];
    __shared__ float s_dir[kRdBlock * 3];
    const long long total = (long long)N * H * W;
    const long long HW = (long long)H * W;
    for (long long base = (long long)blockIdx.x * kRdBlock; base < total; base += (long long)gridDim.x * kRdBlock) {
        const long long r = base + threadIdx.x;
        const bool valid = r < total;
        f3 o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 1.f);
        if (valid) {
            int n, hw;
            if (total < 0x7fffffffll) {  // (uniform) 32-bit division: a 64-bit one is ~80 VALU instructions per wave
                n = (int)((unsigned)r / (unsigned)HW);
                hw = (int)((unsigned)r - (unsigned)n * (unsigned)HW);
            } else {
                n = (int)(r / HW);
                hw = (int)(r - (long long)n * HW);
            }
            const int h = (int)((unsigned)hw / (unsigned)W), w = hw - h * W;
            float px = (float)w, py = (float)h;
            if (pixelcoords) {
                const float2 pc = reinterpret_cast<const float2 *>(pixelcoords)[r];
                px = pc.x;
                py = pc.y;
            }
            CamRay c;
            if ((HW & 63) == 0) {
                // (kernel-uniform) an image is a whole number of waves: the 64 rays of a wave share their camera, whose 19 floats
                // then come through scalar loads instead of 19 vector loads per ray
                const int nu = __builtin_amdgcn_readfirstlane(n);
                const float *cp = campos + nu * 3, *cr = camrot + nu * 9, *cf = focal + nu * 2, *cc = princpt + nu * 2;
                const float Rm[9] = {cload(cr), cload(cr + 1), cload(cr + 2), cload(cr + 3), cload(cr + 4),
                                     cload(cr + 5), cload(cr + 6), cload(cr + 7), cload(cr + 8)};
                c = ray_from_camera(mk3(cload(cp), cload(cp + 1), cload(cp + 2)), Rm, cload(cf), cload(cf + 1), cload(cc),
                                    cload(cc + 1), px, py, volradius);
            } else {
                c = ray_from_camera(ld3(campos + n * 3), camrot + n * 9, focal[n * 2 + 0], focal[n * 2 + 1],
                                    princpt[n * 2 + 0], princpt[n * 2 + 1], px, py, volradius);
            }
            o = c.o;
            d = c.d;
            reinterpret_cast<float2 *>(tminmax)[r] = make_float2(c.tmin, c.tmax);
        }
        st3(s_pos + threadIdx.x * 3, o);
        st3(s_dir + threadIdx.x * 3, d);
        __syncthreads();
        // 256 rays * 3 floats = 768 floats = 192 float4 per stream
        const long long rem = total - base;
        if (rem >= kRdBlock && vec_ok) {  // base*3 floats is a multiple of 768: 16-B aligned iff the arrays are
            if (threadIdx.x < 192) {
                reinterpret_cast<float4 *>(raypos + base * 3)[threadIdx.x] =
                    reinterpret_cast<const float4 *>(s_pos)[threadIdx.x];
                reinterpret_cast<float4 *>(raydir + base * 3)[threadIdx.x] =
                    reinterpret_cast<const float4 *>(s_dir)[threadIdx.x];
            }
        } else {
            const int nflt = (int)(rem < kRdBlock ? rem : kRdBlock) * 3;
            for (int i = threadIdx.x; i < nflt; i += kRdBlock) {
                raypos[base * 3 + i] = s_pos[i];
                raydir[base * 3 + i] = s_dir[i];
            }
        }
        __syncthreads();
    }
}

}  // namespace mvp

extern "C" int mvp_raydirs_forward(int N, int H, int W, const float *campos, const float *camrot,
                                   const float *focal, const float *princpt, const float *pixelcoords,
                                   float volradius, float *raypos, float *raydir, float *tminmax, void *stream) {
    if (N < 0 || H < 0 || W < 0) return MVP_ERR_BADARG;
    if ((long long)N * H * W == 0) return MVP_OK;
    if (!campos || !camrot || !focal || !princpt || !raypos || !raydir || !tminmax) return MVP_ERR_BADARG;
    if (!(volradius > 0.f) || !(volradius < INFINITY)) return MVP_ERR_BADARG;
    const long long total = (long long)N * H * W;
    long long blocks = (total + mvp::kRdBlock - 1) / mvp::kRdBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;  // 256 CUs x 16 resident blocks, grid-stride beyond that
    hipLaunchKernelGGL(mvp::raydirs_kernel, dim3((unsigned)blocks), dim3(mvp::kRdBlock), 0, (hipStream_t)stream, N, H,
                       W, campos, camrot, focal, princpt, pixelcoords, volradius, raypos, raydir, tminmax,
                       (mvp::aligned16(raypos) && mvp::aligned16(raydir)) ? 1 : 0);
    return mvp::launch_status();
}
