// gradclip.hip -- row N4 of SURVEY.md 8(f): the gradient hygiene of the reference's optimisation loop as two
// multi-tensor HIP passes instead of ~4 eager ops + host syncs per parameter tensor.
//
// Reference (ddp-train.py:434-441):
//     for p in params:  p.grad.data[torch.isnan(p.grad.data)] = 0 ;  p.grad.data[torch.isinf(p.grad.data)] = 0
//     torch.nn.utils.clip_grad_norm_(model.parameters(), train_params.clip)
// torch.nn.utils.clip_grad_norm_ (PyTorch, norm_type = 2; third-party to the reference, restated from its published
// algorithm):  total_norm = || ( ||g_1||_2, ..., ||g_n||_2 ) ||_2 ;  coef = max_norm / (total_norm + 1e-6) ;
//     coef = min(coef, 1) ;  g_i *= coef  for every i.
//
// Pass 1 (gc_sanitize_sqnorm_kernel): one streaming read of every gradient; non-finite elements are overwritten
// with 0 (only those 16-byte groups are written back); sum of squares per block in fp32 over <= 64 elements per
// thread, then fp64 across the block and one global_atomic_add_f64 per block.
// Pass 2 (gc_scale_kernel): coef is computed ON THE DEVICE from the fp64 sum (no host round trip); when coef >= 1
// the kernel returns before touching the gradients (x * 1.0f is the identity, so this is what torch produces too).
// HBM-bound: 4 B/element read (pass 1) + 8 B/element (pass 2, only when clipping).
//
// Tensor metadata (pointer, element count, first workgroup) travels in the kernel argument, kGcTensors tensors per
// launch, so the library needs no device-side table and allocates nothing; the grid holds exactly the 16384-float
// chunks that exist (4 launches per pass for ava-256's ~600 parameter tensors).
#include "mvp_device.h"
#include "mvp_host.h"

namespace mvp {

constexpr int kGcTensors = 160;  // per launch: 160 * 20 B of kernel argument (the limit is 4 KB)
constexpr int kGcChunk = 16384;  // floats per workgroup (65536 was tried: fewer fp64 atomics, but 17 -> 26 us per launch)
constexpr int kGcBlock = 256;

struct GcBatch {
    float *ptr[kGcTensors];
    unsigned long long n[kGcTensors];
    unsigned first[kGcTensors];  // index of the tensor's first workgroup in this launch (ascending)
    int count;
};

// workgroup -> (tensor, chunk): the grid holds exactly the chunks that exist; wave-uniform binary search
__device__ __forceinline__ int gc_locate(const GcBatch &b, unsigned blk, unsigned long long &start) {
    int lo = 0, hi = b.count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (b.first[mid] <= blk) lo = mid; else hi = mid - 1;
    }
    start = (unsigned long long)(blk - b.first[lo]) * kGcChunk;
    return lo;
}

__device__ __forceinline__ bool finite_f(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }

__global__ __launch_bounds__(kGcBlock) void gc_sanitize_sqnorm_kernel(const GcBatch b, double *__restrict__ sq) {
    unsigned long long start;
    const int t = gc_locate(b, blockIdx.x, start);
    const unsigned long long n = b.n[t];
    float *g = b.ptr[t] + start;
    const int cnt = (int)min((unsigned long long)kGcChunk, n - start);
    float acc = 0.f;
    if ((reinterpret_cast<uintptr_t>(g) & 15u) == 0) {
        float4 *g4 = reinterpret_cast<float4 *>(g);
        const int n4 = cnt >> 2;
        float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);  // four independent fp32 partial sums of <= 16 terms each
        for (int i = threadIdx.x; i < n4; i += kGcBlock) {
            float4 v = g4[i];
            const bool ok = finite_f(v.x) && finite_f(v.y) && finite_f(v.z) && finite_f(v.w);
            if (!ok) {
                v.x = finite_f(v.x) ? v.x : 0.f, v.y = finite_f(v.y) ? v.y : 0.f;
                v.z = finite_f(v.z) ? v.z : 0.f, v.w = finite_f(v.w) ? v.w : 0.f;
                g4[i] = v;
            }
            a4.x += v.x * v.x, a4.y += v.y * v.y, a4.z += v.z * v.z, a4.w += v.w * v.w;
        }
        acc = (a4.x + a4.y) + (a4.z + a4.w);
        for (int i = (n4 << 2) + threadIdx.x; i < cnt; i += kGcBlock) {
            float v = g[i];
            if (!finite_f(v)) g[i] = v = 0.f;
            acc += v * v;
        }
    } else {  // a view that does not start on a 16-byte boundary
        for (int i = threadIdx.x; i < cnt; i += kGcBlock) {
            float v = g[i];
            if (!finite_f(v)) g[i] = v = 0.f;
            acc += v * v;
        }
    }
    __shared__ double s_part[kGcBlock / 64];
    double d = (double)wave_sum(acc);  // fp32 within a wave (64 lanes x 64 terms), then fp64
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        d = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (d != 0.0) atomicAdd(sq, d);
    }
}

__device__ __forceinline__ float gc_coef(double sq, float max_norm, float *total) {
    const float tn = (float)sqrt(sq);
    *total = tn;
    const float c = max_norm / (tn + 1.0e-6f);
    return c < 1.0f ? c : 1.0f;
}

__global__ __launch_bounds__(kGcBlock) void gc_scale_kernel(const GcBatch b, const double *__restrict__ sq,
                                                            float max_norm, float *__restrict__ total_out) {
    float tn;
    const float coef = gc_coef(*sq, max_norm, &tn);
    if (total_out && blockIdx.x == 0 && threadIdx.x == 0) *total_out = tn;
    if (!(coef < 1.0f)) return;
    unsigned long long start;
    const int t = gc_locate(b, blockIdx.x, start);
    const unsigned long long n = b.n[t];
    float *g = b.ptr[t] + start;
    const int cnt = (int)min((unsigned long long)kGcChunk, n - start);
    if ((reinterpret_cast<uintptr_t>(g) & 15u) == 0) {
        float4 *g4 = reinterpret_cast<float4 *>(g);
        const int n4 = cnt >> 2;
        for (int i = threadIdx.x; i < n4; i += kGcBlock) {
            float4 v = g4[i];
            v.x *= coef, v.y *= coef, v.z *= coef, v.w *= coef;
            g4[i] = v;
        }
        for (int i = (n4 << 2) + threadIdx.x; i < cnt; i += kGcBlock) g[i] *= coef;
    } else {
        for (int i = threadIdx.x; i < cnt; i += kGcBlock) g[i] *= coef;
    }
}

// fills one launch's metadata from the host arrays (empty tensors are skipped); returns its number of workgroups and
// advances `next` past the tensors it took
static unsigned fill_batch(GcBatch &b, int &next, int ntensors, float *const *grads, const long long *numels) {
    unsigned blocks = 0;
    b.count = 0;
    while (next < ntensors && b.count < kGcTensors) {
        const unsigned long long n = (unsigned long long)numels[next];
        if (n > 0) {
            const unsigned long long c = (n + kGcChunk - 1) / kGcChunk;
            if ((unsigned long long)blocks + c > 0x7fffffffull) break;  // next launch
            b.ptr[b.count] = grads[next], b.n[b.count] = n, b.first[b.count] = blocks;
            blocks += (unsigned)c;
            ++b.count;
        }
        ++next;
    }
    for (int j = b.count; j < kGcTensors; ++j) b.ptr[j] = nullptr, b.n[j] = 0ull, b.first[j] = 0xffffffffu;
    return blocks;
}

static int check_list(int ntensors, float *const *grads, const long long *numels) {
    if (ntensors < 0 || (ntensors > 0 && (!grads || !numels))) return MVP_ERR_BADARG;
    for (int i = 0; i < ntensors; ++i) {
        if (numels[i] < 0 || (numels[i] > 0 && !grads[i])) return MVP_ERR_BADARG;
        if ((reinterpret_cast<uintptr_t>(grads[i]) & 3u) != 0) return MVP_ERR_BADARG;
        if ((unsigned long long)numels[i] > (unsigned long long)kGcChunk * 0x7fffffffull) return MVP_ERR_UNSUPPORTED;
    }
    return MVP_OK;
}

}  // namespace mvp

using namespace mvp;

extern "C" int mvp_grads_sanitize_sqnorm(int ntensors, float *const *grads, const long long *numels, double *sqnorm,
                                         void *stream) {
    int rc = check_list(ntensors, grads, numels);
    if (rc != MVP_OK) return rc;
    if (!sqnorm) return MVP_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(sqnorm, 0, sizeof(double), st);
    if (e != hipSuccess) return (int)e;
    for (int next = 0; next < ntensors;) {
        GcBatch b;
        const unsigned blocks = fill_batch(b, next, ntensors, grads, numels);
        if (blocks == 0) continue;
        hipLaunchKernelGGL(gc_sanitize_sqnorm_kernel, dim3(blocks), dim3(kGcBlock), 0, st, b, sqnorm);
        rc = launch_status();
        if (rc != MVP_OK) return rc;
    }
    return MVP_OK;
}

extern "C" int mvp_grads_clip_scale(int ntensors, float *const *grads, const long long *numels, const double *sqnorm,
                                    float max_norm, float *total_norm, void *stream) {
    int rc = check_list(ntensors, grads, numels);
    if (rc != MVP_OK) return rc;
    if (!sqnorm || !(max_norm >= 0.f)) return MVP_ERR_BADARG;
    hipStream_t st = (hipStream_t)stream;
    bool launched = false;
    for (int next = 0; next < ntensors;) {
        GcBatch b;
        const unsigned blocks = fill_batch(b, next, ntensors, grads, numels);
        if (blocks == 0) continue;
        hipLaunchKernelGGL(gc_scale_kernel, dim3(blocks), dim3(kGcBlock), 0, st, b, sqnorm, max_norm,
                           launched ? nullptr : total_norm);
        launched = true;
        rc = launch_status();
        if (rc != MVP_OK) return rc;
    }
    if (!launched && total_norm) {  // nothing to scale: the norm of an empty set is 0
        hipError_t e = hipMemsetAsync(total_norm, 0, sizeof(float), st);
        if (e != hipSuccess) return (int)e;
    }
    return MVP_OK;
}
