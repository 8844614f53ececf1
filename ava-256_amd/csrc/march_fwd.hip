// march_fwd.hip -- C-ABI entry points of the forward march (include/mvp_abi.h: mvp_march_forward,
// mvp_march_forward_cams, mvp_march_render_half) and the instantiations of march_kernel<false, ..> they launch.
//   /root/reference/extensions/mvpraymarch/mvpraymarch.cpp:38-66 (raymarch_forward_cuda), mvpraymarch_kernel.cu:35-120
#include "march_packet.h"

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
                              int primlist_cap, float fadescale, float fadeexp, uint32_t *diag, void *stream,
                              bool half_slabs = false) {
    using namespace mvp;
    MarchParams p = {};
    if (half_slabs) {  // the opt-in render path over fp16 RGBA slabs: forward only, nothing handed to a backward, 8^3 slabs
        if (warp || raysat || rayaux || primlist_count || primlist) return MVP_ERR_BADARG;
        if (TD != 8 || TH != 8 || TW != 8 || (unsigned long long)K * 4096ull >= (1ull << 32)) return MVP_ERR_UNSUPPORTED;
    }
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
    p.rayaux = rayaux, p.pl_count = primlist_count, p.pl_list = reinterpret_cast<uint4 *>(primlist);
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
    const dim3 grid((unsigned)p.total_packets), block(kWave);
    if (warp) {
        if (fade8)
            hipLaunchKernelGGL((march_kernel<false, true, true>), grid, block, 0, st, p);
        else
            hipLaunchKernelGGL((march_kernel<false, false, true>), grid, block, 0, st, p);
    } else if (half_slabs) {
        if (fade8)
            hipLaunchKernelGGL((march_half_kernel<true>), grid, block, 0, st, p);
        else
            hipLaunchKernelGGL((march_half_kernel<false>), grid, block, 0, st, p);
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

// The opt-in render path: the forward march over HALF-PRECISION slabs (fp16 RGBA, 8 bytes per voxel; made by
// mvp_template_to_half or mvp_template_assemble_forward_half).  Rays either as tensors (campos == NULL) or made inside the
// march from the cameras (raypos == raydir == tminmax == NULL), like the two fp32 entry points.  Forward only: no raysat,
// no hand-off -- a training step uses the fp32 path.  8^3 slabs.
extern "C" int mvp_march_render_half(int N, int H, int W, int K, const float *raypos, const float *raydir,
                                     const float *tminmax, const float *campos, const float *camrot, const float *focal,
                                     const float *princpt, const float *pixelcoords, float volradius, float stepsize,
                                     const float *nodeaabb, const float *primpos, const float *primrot,
                                     const float *primscale, int TD, int TH, int TW, const void *tplate_half,
                                     float *rayrgba, float fadescale, float fadeexp, uint32_t *diag, void *stream) {
    if (K > 0 && (!tplate_half || ((uintptr_t)tplate_half & 15u))) return MVP_ERR_BADARG;
    const float *tp = reinterpret_cast<const float *>(tplate_half);  // (MarchParams carries one slab pointer)
    if (campos) {
        if (raypos || raydir || tminmax) return MVP_ERR_BADARG;
        const CameraArgs cams = {campos, camrot, focal, princpt, pixelcoords, volradius, nullptr, nullptr, nullptr};
        return march_forward_impl(N, H, W, K, nullptr, nullptr, &cams, stepsize, nullptr, nodeaabb, primpos, primrot,
                                  primscale, TD, TH, TW, tp, 0, 0, 0, nullptr, rayrgba, nullptr, nullptr, nullptr, nullptr, 0,
                                  fadescale, fadeexp, diag, stream, true);
    }
    return march_forward_impl(N, H, W, K, raypos, raydir, nullptr, stepsize, tminmax, nodeaabb, primpos, primrot, primscale,
                              TD, TH, TW, tp, 0, 0, 0, nullptr, rayrgba, nullptr, nullptr, nullptr, nullptr, 0, fadescale,
                              fadeexp, diag, stream, true);
}
