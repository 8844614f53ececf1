// placement.hip -- the barycentric "postex" half of row N2: where the primitives sit on the mesh.
//
// Reference (models/decoders/assembler.py:118-122): a full 1024 x 1024 x 3 position map per batch element,
//     postex = (bar0 * geo[idx0] + bar1 * geo[idx1] + bar2 * geo[idx2]).permute(0,3,1,2) / volradius
// built with three index_selects of [B, 1048576, 3] (and their scatter-add backward), of which the assembler then
// reads only the texel at each primitive's centre and its +u / +v neighbours (assembler.py:143-206, the two branches
// that work: 256 and 16384 primitives):
//     primpos   = postex[:, :, y0::sy, x0::sx]                       (centre texel c = (y0 + i*sy, x0 + j*sx))
//     vcenterdu = (postex[..., 1:] - postex[..., :-1])[..., c]       = postex(c + (0,1)) - postex(c)
//     vcenterdv = (postex[:, :, 1:] - postex[:, :, :-1])[..., c]     = postex(c + (1,0)) - postex(c)
// i.e. 3 of every sy*sx texels (1.2 % of the map at 16384 primitives).  Here one thread computes exactly those three
// texels of one (batch element, primitive) -- 9 vertex fetches -- with the reference's operation order, so the three
// outputs are BIT-IDENTICAL to the eager expression (fp contraction off).  The backward scatters into grad_geo with
// fp32 atomics (B*K*27 of them, spread over ~7k vertices x 3 floats), which is also what the reference's index_add
// does -- compared with a tolerance.
//
// Layouts: geo [B, V, 3], idxim [T, T, 3] int32 (vertex indices), barim [T, T, 3] float32, outputs [B, K, 3] with
// k = i * nx + j (row-major over the centre grid, the order of `.view(B, nprims, 3)` at assembler.py:143).
#include "mvp_device.h"
#include "mvp_host.h"

namespace mvp {

struct PlaceParams {
    int B, V, T, ny, nx, y0, sy, x0, sx;
    float volradius;
    const float *geo;
    const int *idxim;
    const float *barim;
};

#pragma clang fp contract(off)
__device__ __forceinline__ f3 postex_at(const PlaceParams &p, const float *gb, int y, int x) {
    const size_t t = ((size_t)y * p.T + x) * 3;
    const int i0 = p.idxim[t], i1 = p.idxim[t + 1], i2 = p.idxim[t + 2];
    const float b0 = p.barim[t], b1 = p.barim[t + 1], b2 = p.barim[t + 2];
    const f3 v0 = ld3(gb + (size_t)i0 * 3), v1 = ld3(gb + (size_t)i1 * 3), v2 = ld3(gb + (size_t)i2 * 3);
    // (bar0 * g0 + bar1 * g1) + bar2 * g2, then / volradius: the eager expression's order, one rounding per operation
    return mk3(((b0 * v0.x + b1 * v1.x) + b2 * v2.x) / p.volradius, ((b0 * v0.y + b1 * v1.y) + b2 * v2.y) / p.volradius,
               ((b0 * v0.z + b1 * v1.z) + b2 * v2.z) / p.volradius);
}

__global__ __launch_bounds__(256) void placement_fwd_kernel(const PlaceParams p, float *__restrict__ primpos,
                                                            float *__restrict__ du, float *__restrict__ dv) {
    const int K = p.ny * p.nx;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (size_t)p.B * K) return;
    const int b = (int)(g / K), k = (int)(g - (size_t)b * K);
    const int i = k / p.nx, j = k - i * p.nx;
    const int y = p.y0 + i * p.sy, x = p.x0 + j * p.sx;
    const float *gb = p.geo + (size_t)b * p.V * 3;
    const f3 c = postex_at(p, gb, y, x), cu = postex_at(p, gb, y, x + 1), cv = postex_at(p, gb, y + 1, x);
    st3(primpos + g * 3, c);
    st3(du + g * 3, mk3(cu.x - c.x, cu.y - c.y, cu.z - c.z));
    st3(dv + g * 3, mk3(cv.x - c.x, cv.y - c.y, cv.z - c.z));
}

__device__ __forceinline__ void scatter_texel(const PlaceParams &p, float *ggb, int y, int x, f3 g) {
    const size_t t = ((size_t)y * p.T + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int i = p.idxim[t + c];
        const float w = p.barim[t + c];
        float *dst = ggb + (size_t)i * 3;
        atomicAdd(dst + 0, w * g.x);
        atomicAdd(dst + 1, w * g.y);
        atomicAdd(dst + 2, w * g.z);
    }
}

__global__ __launch_bounds__(256) void placement_bwd_kernel(const PlaceParams p, const float *__restrict__ g_pos,
                                                            const float *__restrict__ g_du,
                                                            const float *__restrict__ g_dv, float *grad_geo) {
    const int K = p.ny * p.nx;
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (size_t)p.B * K) return;
    const int b = (int)(g / K), k = (int)(g - (size_t)b * K);
    const int i = k / p.nx, j = k - i * p.nx;
    const int y = p.y0 + i * p.sy, x = p.x0 + j * p.sx;
    float *ggb = grad_geo + (size_t)b * p.V * 3;
    const float inv = 1.0f / p.volradius;
    const f3 gp = g_pos ? ld3(g_pos + g * 3) : mk3(0.f, 0.f, 0.f);
    const f3 gu = g_du ? ld3(g_du + g * 3) : mk3(0.f, 0.f, 0.f);
    const f3 gv = g_dv ? ld3(g_dv + g * 3) : mk3(0.f, 0.f, 0.f);
    // d/d postex(c) = g_pos - g_du - g_dv ;  d/d postex(c+u) = g_du ;  d/d postex(c+v) = g_dv ;  postex = (...) / volradius
    scatter_texel(p, ggb, y, x, mk3((gp.x - gu.x - gv.x) * inv, (gp.y - gu.y - gv.y) * inv, (gp.z - gu.z - gv.z) * inv));
    scatter_texel(p, ggb, y, x + 1, mk3(gu.x * inv, gu.y * inv, gu.z * inv));
    scatter_texel(p, ggb, y + 1, x, mk3(gv.x * inv, gv.y * inv, gv.z * inv));
}

static int fill_params(PlaceParams &p, int B, int V, int T, int ny, int nx, int y0, int sy, int x0, int sx,
                       float volradius, const float *geo, const int *idxim, const float *barim) {
    if (B < 0 || V <= 0 || T <= 1 || ny <= 0 || nx <= 0 || y0 < 0 || x0 < 0 || sy <= 0 || sx <= 0) return MVP_ERR_BADARG;
    if (!(volradius > 0.f) || !geo || !idxim || !barim) return MVP_ERR_BADARG;
    // the +u / +v neighbours of the last centre must exist
    if ((long long)y0 + (long long)(ny - 1) * sy + 1 >= T || (long long)x0 + (long long)(nx - 1) * sx + 1 >= T)
        return MVP_ERR_BADARG;
    if ((long long)B * ny * nx > 0x7fffffffll * 256) return MVP_ERR_UNSUPPORTED;
    p.B = B, p.V = V, p.T = T, p.ny = ny, p.nx = nx, p.y0 = y0, p.sy = sy, p.x0 = x0, p.sx = sx;
    p.volradius = volradius, p.geo = geo, p.idxim = idxim, p.barim = barim;
    return MVP_OK;
}

}  // namespace mvp

using namespace mvp;

extern "C" int mvp_prim_placement_forward(int B, int V, int T, int ny, int nx, int y0, int sy, int x0, int sx,
                                          float volradius, const float *geo, const int *idxim, const float *barim,
                                          float *primpos, float *vcenterdu, float *vcenterdv, void *stream) {
    PlaceParams p;
    int rc = fill_params(p, B, V, T, ny, nx, y0, sy, x0, sx, volradius, geo, idxim, barim);
    if (rc != MVP_OK) return rc;
    if (!primpos || !vcenterdu || !vcenterdv) return MVP_ERR_BADARG;
    const long long n = (long long)B * ny * nx;
    if (n == 0) return MVP_OK;
    hipLaunchKernelGGL(placement_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p,
                       primpos, vcenterdu, vcenterdv);
    return launch_status();
}

extern "C" int mvp_prim_placement_backward(int B, int V, int T, int ny, int nx, int y0, int sy, int x0, int sx,
                                           float volradius, const int *idxim, const float *barim,
                                           const float *grad_primpos, const float *grad_vcenterdu,
                                           const float *grad_vcenterdv, float *grad_geo, void *stream) {
    PlaceParams p;
    int rc = fill_params(p, B, V, T, ny, nx, y0, sy, x0, sx, volradius, grad_geo, idxim, barim);
    if (rc != MVP_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(grad_geo, 0, sizeof(float) * 3 * (size_t)B * V, st);  // grad_geo is OVERWRITTEN
    if (e != hipSuccess) return (int)e;
    const long long n = (long long)B * ny * nx;
    if (n == 0) return MVP_OK;
    hipLaunchKernelGGL(placement_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, grad_primpos,
                       grad_vcenterdu, grad_vcenterdv, grad_geo);
    return launch_status();
}
