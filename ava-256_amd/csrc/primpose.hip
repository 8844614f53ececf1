// primpose.hip -- the residual half of row N2: how the decoder's per-primitive residuals turn the mesh placement
// into the poses the march takes.
//
// Reference (models/decoders/assembler.py:241-253, eager PyTorch; models/utils.py Rodrigues for `self.rodrig`):
//     rw = clamp(residuals_weight, 0, 1)
//     if rw < 1:  posres *= rw;  rotres *= rw;  scaleres = scaleres * rw + (1 - rw)
//     primpos   = primpos + posres
//     primrot   = bmm(primrot, rodrig(rotres))          rodrig: theta = sqrt(1e-5 + |v|^2), a = v / theta,
//     primscale = primscale * scaleres                           R = cos I + (1 - cos) a a^T + sin [a]x
// ~25 kernels forward and ~60 backward on [N, K, 3] / [N, K, 3, 3] tensors (the per-element Rodrigues expressions alone are
// ~200 statements), among them a batched 3x3 GEMM for which hipBLASLt takes 150 us per 16384 products.  Here: one kernel
// each way.  Every input may be shared by the frames (frame stride 0: the stand-in decoder's residuals are per primitive, the
// reference's are per frame and primitive); the backward sums a shared input's gradient over the frames.
//
// Mapping: 256 threads = 64 consecutive primitives x 4 frame lanes (lane l of primitive k takes frames l, l + 4, ...: wave w
// of the workgroup is frame lane w, so a wave reads 64 consecutive primitives of one frame -- 768 contiguous bytes of a
// [K, 3] array); the four frame lanes' sums of a shared input's gradient meet in LDS.  Data per launch: N * K * 15 floats each
// way (C2: 20 MB, C3: 4 MB): latency-bound, not bandwidth-bound.
#include "mvp_device.h"
#include "mvp_host.h"

namespace mvp {

constexpr int kPoseLanes = 4;   // frame lanes per primitive = waves per workgroup
constexpr int kPoseBlock = 64 * kPoseLanes;

struct PoseIn {
    const float *pos0, *rot0, *scale0, *posres, *rotres, *scaleres;
    long long pos0_sn, rot0_sn, posres_sn, rotres_sn, scaleres_sn;  // frame strides in floats (0 = shared by the frames)
    long long scale0_sn, scale0_sk, scale0_sc;                      // any broadcast shape of the base scale
    int N, K;
    float rw;  // already clamped to [0, 1]
};

struct Rod {  // Rodrigues of one axis-angle vector, with what its backward needs
    float R[9], a[3], theta, c, s;
};
__device__ __forceinline__ Rod rodrigues_of(float vx, float vy, float vz) {
    Rod r;
    r.theta = sqrtf(1e-5f + (vx * vx + vy * vy + vz * vz));
    const float it = 1.0f / r.theta;
    r.a[0] = vx * it, r.a[1] = vy * it, r.a[2] = vz * it;
    r.c = cosf(r.theta), r.s = sinf(r.theta);
    const float x = r.a[0], y = r.a[1], z = r.a[2], omc = 1.0f - r.c;
    r.R[0] = x * x + (1.0f - x * x) * r.c, r.R[1] = x * y * omc - z * r.s, r.R[2] = x * z * omc + y * r.s;
    r.R[3] = x * y * omc + z * r.s, r.R[4] = y * y + (1.0f - y * y) * r.c, r.R[5] = y * z * omc - x * r.s;
    r.R[6] = x * z * omc - y * r.s, r.R[7] = y * z * omc + x * r.s, r.R[8] = z * z + (1.0f - z * z) * r.c;
    return r;
}

__global__ __launch_bounds__(kPoseBlock) void pose_fwd_kernel(const PoseIn p, float *__restrict__ primpos,
                                                              float *__restrict__ primrot, float *__restrict__ primscale) {
    const int k = blockIdx.x * 64 + (threadIdx.x & 63), lane_n = threadIdx.x >> 6;
    if (k >= p.K) return;
    const float rw = p.rw, orw = 1.0f - p.rw;
    const bool blend = rw < 1.0f;
    for (int n = lane_n; n < p.N; n += kPoseLanes) {
        const size_t o = (size_t)n * p.K + k;
        f3 pr = ld3(p.posres + n * p.posres_sn + (size_t)k * 3);
        f3 rr = ld3(p.rotres + n * p.rotres_sn + (size_t)k * 3);
        f3 sr = ld3(p.scaleres + n * p.scaleres_sn + (size_t)k * 3);
        if (blend) {  // assembler.py:242-245
            pr = mk3(pr.x * rw, pr.y * rw, pr.z * rw), rr = mk3(rr.x * rw, rr.y * rw, rr.z * rw);
            sr = mk3(sr.x * rw + orw, sr.y * rw + orw, sr.z * rw + orw);
        }
        const f3 p0 = ld3(p.pos0 + n * p.pos0_sn + (size_t)k * 3);
        st3(primpos + o * 3, mk3(p0.x + pr.x, p0.y + pr.y, p0.z + pr.z));
        const float *s0 = p.scale0 + n * p.scale0_sn + k * p.scale0_sk;
        st3(primscale + o * 3, mk3(s0[0] * sr.x, s0[p.scale0_sc] * sr.y, s0[2 * p.scale0_sc] * sr.z));
        const Rod q = rodrigues_of(rr.x, rr.y, rr.z);
        const float *A = p.rot0 + n * p.rot0_sn + (size_t)k * 9;
        float *out = primrot + o * 9;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) out[i * 3 + j] = A[i * 3] * q.R[j] + A[i * 3 + 1] * q.R[3 + j] + A[i * 3 + 2] * q.R[6 + j];
    }
}

struct PoseGrad {
    const float *g_pos, *g_rot, *g_scale;                            // [N, K, 3], [N, K, 3, 3], [N, K, 3]
    float *g_pos0, *g_rot0, *g_posres, *g_rotres, *g_scaleres;       // shaped like their inputs; g_pos0 / g_rot0 may be NULL
};

// One thread's 3 or 9 gradient floats of one input: written per frame when the input is per frame, otherwise summed over the
// thread's frames and, at the end, over the workgroup's four frame lanes.
template <int C>
struct Acc {
    float v[C];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < C; ++i) v[i] = 0.f;
    }
};
template <int C>
__device__ __forceinline__ void emit(float *dst, long long sn, int n, int k, const float (&g)[C], Acc<C> &acc) {
    if (!dst) return;
    if (sn != 0) {
#pragma unroll
        for (int i = 0; i < C; ++i) dst[n * sn + (size_t)k * C + i] = g[i];
    } else {
#pragma unroll
        for (int i = 0; i < C; ++i) acc.v[i] += g[i];
    }
}
template <int C>
__device__ __forceinline__ void flush(float *dst, long long sn, int k, bool live, const Acc<C> &acc, float *s_red) {
    if (!dst || sn != 0) return;  // (workgroup-uniform)
    const int kl = threadIdx.x & 63, lane_n = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < C; ++i) s_red[(lane_n * C + i) * 64 + kl] = acc.v[i];
    __syncthreads();
    if (lane_n == 0 && live) {
#pragma unroll
        for (int i = 0; i < C; ++i) {
            float t = 0.f;
#pragma unroll
            for (int l = 0; l < kPoseLanes; ++l) t += s_red[(l * C + i) * 64 + kl];
            dst[(size_t)k * C + i] = t;
        }
    }
}

__global__ __launch_bounds__(kPoseBlock) void pose_bwd_kernel(const PoseIn p, const PoseGrad q) {
    __shared__ float s_red[kPoseLanes * 9 * 64];
    const int k = blockIdx.x * 64 + (threadIdx.x & 63), lane_n = threadIdx.x >> 6;
    const bool live = k < p.K;
    const float rw = p.rw;
    const bool blend = rw < 1.0f;
    Acc<3> a_pos0, a_posres, a_rotres, a_scaleres;
    Acc<9> a_rot0;
    a_pos0.zero(), a_posres.zero(), a_rotres.zero(), a_scaleres.zero(), a_rot0.zero();
    for (int n = lane_n; live && n < p.N; n += kPoseLanes) {
        const size_t o = (size_t)n * p.K + k;
        // ---- position: primpos = pos0 + posres * rw ----
        const f3 gp = ld3(q.g_pos + o * 3);
        const float gpos[3] = {gp.x, gp.y, gp.z};
        const float gpr[3] = {blend ? gp.x * rw : gp.x, blend ? gp.y * rw : gp.y, blend ? gp.z * rw : gp.z};
        emit<3>(q.g_pos0, p.pos0_sn, n, k, gpos, a_pos0);
        emit<3>(q.g_posres, p.posres_sn, n, k, gpr, a_posres);
        // ---- scale: primscale = scale0 * (scaleres * rw + 1 - rw) ----
        const f3 gs = ld3(q.g_scale + o * 3);
        const float *s0 = p.scale0 + n * p.scale0_sn + k * p.scale0_sk;
        const float m = blend ? rw : 1.0f;
        const float gsr[3] = {gs.x * s0[0] * m, gs.y * s0[p.scale0_sc] * m, gs.z * s0[2 * p.scale0_sc] * m};
        emit<3>(q.g_scaleres, p.scaleres_sn, n, k, gsr, a_scaleres);
        // ---- rotation: primrot = A Rres(v), v = rotres * rw ----
        f3 rr = ld3(p.rotres + n * p.rotres_sn + (size_t)k * 3);
        if (blend) rr = mk3(rr.x * rw, rr.y * rw, rr.z * rw);
        const Rod r = rodrigues_of(rr.x, rr.y, rr.z);
        const float *A = p.rot0 + n * p.rot0_sn + (size_t)k * 9;
        const float *Go = q.g_rot + o * 9;
        float Gv[9], a9[9], G[9], GA[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) Gv[i] = Go[i], a9[i] = A[i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                G[i * 3 + j] = a9[i] * Gv[j] + a9[3 + i] * Gv[3 + j] + a9[6 + i] * Gv[6 + j];             // A^T Gout
                GA[i * 3 + j] = Gv[i * 3] * r.R[j * 3] + Gv[i * 3 + 1] * r.R[j * 3 + 1] + Gv[i * 3 + 2] * r.R[j * 3 + 2];  // Gout R^T
            }
        emit<9>(q.g_rot0, p.rot0_sn, n, k, GA, a_rot0);
        // R = c I + (1 - c) a a^T + s [a]x (diagonal: a_i^2 + (1 - a_i^2) c, the same thing)
        const float ax = r.a[0], ay = r.a[1], az = r.a[2], omc = 1.0f - r.c;
        const float aGa = ax * (G[0] * ax + G[1] * ay + G[2] * az) + ay * (G[3] * ax + G[4] * ay + G[5] * az) +
                          az * (G[6] * ax + G[7] * ay + G[8] * az);
        const float wx = G[7] - G[5], wy = G[2] - G[6], wz = G[3] - G[1];      // dR/ds = [a]x  ->  <G, [a]x> = a . w
        const float dLdc = (G[0] + G[4] + G[8]) - aGa, dLds = ax * wx + ay * wy + az * wz;
        // dL/da = (1 - c) (G + G^T) a + s w
        const float dax = omc * (2.0f * G[0] * ax + (G[1] + G[3]) * ay + (G[2] + G[6]) * az) + r.s * wx;
        const float day = omc * ((G[1] + G[3]) * ax + 2.0f * G[4] * ay + (G[5] + G[7]) * az) + r.s * wy;
        const float daz = omc * ((G[2] + G[6]) * ax + (G[5] + G[7]) * ay + 2.0f * G[8] * az) + r.s * wz;
        const float it = 1.0f / r.theta;
        // theta enters through cos / sin and through a = v / theta; d theta / d v = a
        const float dLdth = -r.s * dLdc + r.c * dLds - (dax * ax + day * ay + daz * az) * it;
        const float grv[3] = {(dax * it + dLdth * ax) * m, (day * it + dLdth * ay) * m, (daz * it + dLdth * az) * m};
        emit<3>(q.g_rotres, p.rotres_sn, n, k, grv, a_rotres);
    }
    flush<3>(q.g_pos0, p.pos0_sn, k, live, a_pos0, s_red);
    flush<3>(q.g_posres, p.posres_sn, k, live, a_posres, s_red);
    flush<3>(q.g_scaleres, p.scaleres_sn, k, live, a_scaleres, s_red);
    flush<3>(q.g_rotres, p.rotres_sn, k, live, a_rotres, s_red);
    flush<9>(q.g_rot0, p.rot0_sn, k, live, a_rot0, s_red);
}

// ---- the TBN frame (assembler.py:226-239): primrot0 from the centre texel's +u / +v differences --------------------------
//     tangent = du / max(|du|, 1e-8);  normal = cross(tangent, dv), normalised the same way;  bitangent = cross(normal,
//     tangent), normalised;  primrot = stack((tangent, bitangent, normal), dim=-2).permute(.., 3, 2): COLUMNS t, b, n.
// Eager: ~15 kernels forward, ~45 backward on [B, K, 3] tensors.  One thread per (frame, primitive).
__device__ __forceinline__ f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// x / clamp(|x|, min = 1e-8): returns the divisor too
__device__ __forceinline__ f3 unit_clamped(f3 x, float &m) {
    m = fmaxf(sqrtf(dot3(x, x)), 1e-8f);
    return mk3(x.x / m, x.y / m, x.z / m);
}
// gradient through y = x / clamp(|x|, 1e-8): (g - y (y . g)) / m above the clamp, g / 1e-8 on it (the clamp's gradient is 0)
__device__ __forceinline__ f3 unit_clamped_bwd(f3 y, float m, f3 g) {
    if (m > 1e-8f) {
        const float yg = dot3(y, g);
        return mk3((g.x - y.x * yg) / m, (g.y - y.y * yg) / m, (g.z - y.z * yg) / m);
    }
    return mk3(g.x / m, g.y / m, g.z / m);
}

__global__ __launch_bounds__(256) void frame_fwd_kernel(int M, const float *__restrict__ du, const float *__restrict__ dv,
                                                        float *__restrict__ primrot) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    float mt, mn, mb;
    const f3 t = unit_clamped(ld3(du + (size_t)i * 3), mt);
    const f3 n = unit_clamped(cross3(t, ld3(dv + (size_t)i * 3)), mn);
    const f3 b = unit_clamped(cross3(n, t), mb);
    float *R = primrot + (size_t)i * 9;
    R[0] = t.x, R[1] = b.x, R[2] = n.x;
    R[3] = t.y, R[4] = b.y, R[5] = n.y;
    R[6] = t.z, R[7] = b.z, R[8] = n.z;
}

__global__ __launch_bounds__(256) void frame_bwd_kernel(int M, const float *__restrict__ du, const float *__restrict__ dv,
                                                        const float *__restrict__ g_rot, float *__restrict__ g_du,
                                                        float *__restrict__ g_dv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    float mt, mn, mb;
    const f3 v = ld3(dv + (size_t)i * 3);
    const f3 t = unit_clamped(ld3(du + (size_t)i * 3), mt);
    const f3 n = unit_clamped(cross3(t, v), mn);
    const f3 b = unit_clamped(cross3(n, t), mb);
    const float *G = g_rot + (size_t)i * 9;
    f3 gt = mk3(G[0], G[3], G[6]);
    const f3 gb = mk3(G[1], G[4], G[7]);
    f3 gn = mk3(G[2], G[5], G[8]);
    // b = unit(n x t):  d/dn = t x g_b0,  d/dt = g_b0 x n
    const f3 gb0 = unit_clamped_bwd(b, mb, gb);
    const f3 c1 = cross3(t, gb0), c2 = cross3(gb0, n);
    gn = mk3(gn.x + c1.x, gn.y + c1.y, gn.z + c1.z);
    gt = mk3(gt.x + c2.x, gt.y + c2.y, gt.z + c2.z);
    // n = unit(t x dv):  d/dt = dv x g_n0,  d/d dv = g_n0 x t
    const f3 gn0 = unit_clamped_bwd(n, mn, gn);
    const f3 c3 = cross3(v, gn0);
    gt = mk3(gt.x + c3.x, gt.y + c3.y, gt.z + c3.z);
    st3(g_dv + (size_t)i * 3, cross3(gn0, t));
    st3(g_du + (size_t)i * 3, unit_clamped_bwd(t, mt, gt));
}

static int pose_args_ok(const PoseIn &p) {
    if (p.N < 0 || p.K < 0) return MVP_ERR_BADARG;
    if ((long long)p.N * p.K == 0) return MVP_OK;
    if (!p.pos0 || !p.rot0 || !p.scale0 || !p.posres || !p.rotres || !p.scaleres) return MVP_ERR_BADARG;
    if (!(p.rw >= 0.0f && p.rw <= 1.0f)) return MVP_ERR_BADARG;
    const long long s3 = (long long)p.K * 3, s9 = (long long)p.K * 9;
    if ((p.pos0_sn != 0 && p.pos0_sn != s3) || (p.posres_sn != 0 && p.posres_sn != s3) ||
        (p.rotres_sn != 0 && p.rotres_sn != s3) || (p.scaleres_sn != 0 && p.scaleres_sn != s3) ||
        (p.rot0_sn != 0 && p.rot0_sn != s9))
        return MVP_ERR_BADARG;
    if (p.scale0_sn < 0 || p.scale0_sk < 0 || p.scale0_sc < 0) return MVP_ERR_BADARG;
    if ((long long)p.N * p.K > 0x7fffffffll / 16) return MVP_ERR_UNSUPPORTED;
    return MVP_OK;
}

}  // namespace mvp

#define MVP_POSE_IN                                                                                                   \
    mvp::PoseIn p;                                                                                                    \
    p.pos0 = pos0, p.rot0 = rot0, p.scale0 = scale0, p.posres = posres, p.rotres = rotres, p.scaleres = scaleres;    \
    p.pos0_sn = pos0_sn, p.rot0_sn = rot0_sn, p.posres_sn = posres_sn, p.rotres_sn = rotres_sn;                       \
    p.scaleres_sn = scaleres_sn, p.scale0_sn = scale0_sn, p.scale0_sk = scale0_sk, p.scale0_sc = scale0_sc;           \
    p.N = N, p.K = K, p.rw = rw;

extern "C" int mvp_prim_residuals_forward(int N, int K, float rw, const float *pos0, long long pos0_sn, const float *rot0,
                                          long long rot0_sn, const float *scale0, long long scale0_sn, long long scale0_sk,
                                          long long scale0_sc, const float *posres, long long posres_sn, const float *rotres,
                                          long long rotres_sn, const float *scaleres, long long scaleres_sn, float *primpos,
                                          float *primrot, float *primscale, void *stream) {
    MVP_POSE_IN
    const int rc = mvp::pose_args_ok(p);
    if (rc != MVP_OK) return rc;
    if ((long long)N * K == 0) return MVP_OK;
    if (!primpos || !primrot || !primscale) return MVP_ERR_BADARG;
    hipLaunchKernelGGL(mvp::pose_fwd_kernel, dim3((unsigned)((K + 63) / 64)), dim3(mvp::kPoseBlock), 0, (hipStream_t)stream,
                       p, primpos, primrot, primscale);
    return mvp::launch_status();
}

extern "C" int mvp_prim_residuals_backward(int N, int K, float rw, const float *pos0, long long pos0_sn, const float *rot0,
                                           long long rot0_sn, const float *scale0, long long scale0_sn, long long scale0_sk,
                                           long long scale0_sc, const float *posres, long long posres_sn,
                                           const float *rotres, long long rotres_sn, const float *scaleres,
                                           long long scaleres_sn, const float *grad_primpos, const float *grad_primrot,
                                           const float *grad_primscale, float *grad_pos0 /*or NULL*/,
                                           float *grad_rot0 /*or NULL*/, float *grad_posres, float *grad_rotres,
                                           float *grad_scaleres, void *stream) {
    MVP_POSE_IN
    const int rc = mvp::pose_args_ok(p);
    if (rc != MVP_OK) return rc;
    if (K == 0) return MVP_OK;
    if (N == 0) {  // no frame: the per-frame gradients are empty, the shared ones are zero
        if (grad_pos0 && pos0_sn == 0) (void)hipMemsetAsync(grad_pos0, 0, sizeof(float) * 3 * (size_t)K, (hipStream_t)stream);
        if (grad_rot0 && rot0_sn == 0) (void)hipMemsetAsync(grad_rot0, 0, sizeof(float) * 9 * (size_t)K, (hipStream_t)stream);
        if (grad_posres && posres_sn == 0) (void)hipMemsetAsync(grad_posres, 0, sizeof(float) * 3 * (size_t)K, (hipStream_t)stream);
        if (grad_rotres && rotres_sn == 0) (void)hipMemsetAsync(grad_rotres, 0, sizeof(float) * 3 * (size_t)K, (hipStream_t)stream);
        if (grad_scaleres && scaleres_sn == 0)
            (void)hipMemsetAsync(grad_scaleres, 0, sizeof(float) * 3 * (size_t)K, (hipStream_t)stream);
        return mvp::launch_status();
    }
    if (!grad_primpos || !grad_primrot || !grad_primscale || !grad_posres || !grad_rotres || !grad_scaleres)
        return MVP_ERR_BADARG;
    mvp::PoseGrad q;
    q.g_pos = grad_primpos, q.g_rot = grad_primrot, q.g_scale = grad_primscale;
    q.g_pos0 = grad_pos0, q.g_rot0 = grad_rot0, q.g_posres = grad_posres, q.g_rotres = grad_rotres, q.g_scaleres = grad_scaleres;
    hipLaunchKernelGGL(mvp::pose_bwd_kernel, dim3((unsigned)((K + 63) / 64)), dim3(mvp::kPoseBlock), 0, (hipStream_t)stream, p, q);
    return mvp::launch_status();
}

extern "C" int mvp_prim_frame_forward(long long M, const float *du, const float *dv, float *primrot, void *stream) {
    if (M < 0) return MVP_ERR_BADARG;
    if (M == 0) return MVP_OK;
    if (!du || !dv || !primrot) return MVP_ERR_BADARG;
    if (M > 0x7fffffffll / 16) return MVP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(mvp::frame_fwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)M, du, dv,
                       primrot);
    return mvp::launch_status();
}

extern "C" int mvp_prim_frame_backward(long long M, const float *du, const float *dv, const float *grad_primrot, float *grad_du,
                                       float *grad_dv, void *stream) {
    if (M < 0) return MVP_ERR_BADARG;
    if (M == 0) return MVP_OK;
    if (!du || !dv || !grad_primrot || !grad_du || !grad_dv) return MVP_ERR_BADARG;
    if (M > 0x7fffffffll / 16) return MVP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(mvp::frame_bwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)M, du, dv,
                       grad_primrot, grad_du, grad_dv);
    return mvp::launch_status();
}
