// bgmlp.hip -- row N4 of SURVEY.md 8(f), second half: the per-pixel background MLP of the training loop
// (/root/reference/models/bg/mlp2d.py:29-41,56-72) as two fused MFMA kernels.
//
// Reference: posenc = cat([sin(2^i pi x) for i < 10] + [cos(2^i pi x) for i < 10]) of the pixel's two normalised
// coordinates (40 channels), concatenated with a 40-channel camera code and a 40-channel identity code that are
// constant over an image, then 1x1 convolutions 120 -> 256 -> 256 -> 256 -> 256 -> 256 -> 3 with LeakyReLU(0.2)
// between them, output * 25 + 100.  Per pixel 0.59 MFLOP; 154 GFLOP per 512 x 512 image: the largest dense
// contraction of a training step, and the only genuine GEMM near the raymarch path.  In eager PyTorch (bf16
// autocast) it is 6 GEMMs + ~25 elementwise / reduction kernels each way that move the [pixels, 256] activations
// through HBM ~20 times; GEMM time is one third of the total (profiles/r02z_train_C3_kernel_stats.csv).
//
// Here: one workgroup (8 waves, one per CU) owns a tile of 256 pixels and carries it through ALL layers.  The
// activations of the tile stay in LDS as bf16 ([256][256 + 8], 132 KB: the 16-byte pad makes the MFMA A-fragment reads
// conflict-free); each wave computes a 64 x 128 block of the tile's [256 x 256] output with v_mfma_f32_32x32x16_bf16
// (8 accumulator tiles = 128 registers).  The weights (128 KB per layer, L2-resident, nn.Linear layout [out][in]: the 8
// consecutive input channels a B fragment needs are 16 contiguous bytes) stream through a 3 x 8 KB LDS ring in chunks of
// 16 input channels: every thread fetches 16 bytes of chunk c + 4 into a register while chunk c is multiplied, so an
// L2 round trip is covered by four chunks of MFMA work, and a weight element is read once per 256 pixels.  (A first
// version with 128-pixel tiles whose waves read their B fragments straight from global memory ran at 370 TFLOP/s:
// each k-step waited for an L2 round trip.)  Bias + LeakyReLU are applied on the accumulators, the result goes back to
// LDS as the next layer's input and (training) to HBM once, as bf16, for the backward.  The camera / identity codes enter
// as a per-image bias of the first layer (their 80 input channels are constant over the image), so the first GEMM has
// K = 40 (padded to 48).  The last layer (256 -> 3) is a VALU dot product.
//
// HOW THE PLANES REACH HBM (round 5).  Vector loads and stores retire through ONE in-order counter (vmcnt), so a load
// issued after a store can only be waited for together with that store's acknowledgement.  Rounds 3-4 stored a layer's
// plane as a burst of 16 rows per thread right after the epilogue: the next GEMM's first weight request then waited for
// the whole burst to drain (forward), and the backward's mask pass, which alternated 2 activation loads with 2 gradient
// stores, paid a read latency plus a write acknowledgement per round -- the matrix pipe was busy 28 % / 14 % of the two
// kernels and the waves parked 42 % / 65 % of their time.  Now (a) a plane leaves from inside the GEMM that READS it (the
// tile is constant then), one row per thread and k-step, issued after that step's weight request (tile_gemm, ST); (b) a
// mask pass requests all 16 activation rows before it stores anything; (c) biases are requested before the GEMM and parked
// in LDS, W6 and the next tile's first weight chunks and pixel are requested ahead of the tile's last burst.  Same
// arithmetic, same bytes: at 4 x 512^2 the backward takes 37 % and the training forward 28 % fewer cycles
// (profiles/r05_bgmlp_counters.txt), MFMA busy 0.23 / 0.38, and both move their planes at HBM speed.  Inference is its own
// instantiation (fwd_kernel<false>) with no store code at all.
//
// Backward: the same tile walk in reverse for the INPUT gradients (dZ_l = (dZ_{l+1} . W_{l+1}) * leaky'(A_l)), with
// the transposed weights as B operand; dZ_l is written once as bf16.  Weight and bias gradients are plain
// [256 x P] . [P x 256] GEMMs / column sums over the stored (A_{l-1}, dZ_l) pairs and stay with the BLAS library
// (ava-256_amd/bgmlp.py).
//
// MFMA operand maps (verified on the device: tools/ubench/mfma_probe.hip): A: lane l holds A[l % 32][8 (l / 32) + 0..7];
// B: lane l holds B[8 (l / 32) + 0..7][l % 32]; D: register r of lane l is D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
#include "mvp_device.h"
#include "mvp_host.h"

namespace mvp {
namespace bgmlp {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int kTileM = 256;          // pixels per workgroup tile
constexpr int kWidth = 256;          // hidden width
constexpr int kLdX = kWidth + 8;     // LDS row stride (bf16 elements): 528 bytes
constexpr int kPos = 40;             // positional channels
constexpr int kK0 = 48;              // ... padded to three MFMA k-steps
constexpr int kHidden = 4;           // 256 -> 256 layers
constexpr int kThreads = 512;        // 8 waves: 4 (rows) x 2 (columns) blocks of 64 x 128
constexpr int kChunk = 16;           // input channels per weight chunk = one MFMA k-step
constexpr int kRingElems = kWidth * kChunk;  // bf16 elements per ring buffer (8 KB)
constexpr int kRing = 3;             // ring buffers: one being multiplied, one being read into fragments, one being written
constexpr float kSlope = 0.2f;       // LeakyReLU(0.2), mlp2d.py:30-38
constexpr int kMaskPasses = kTileM * (kWidth / 8) / kThreads;  // 16-byte units of a [tile x 256] bf16 plane per thread: 16

struct Params {
    int B, HW, tiles_per_image;
    const float *samplecoords;  // [B, HW, 2]
    const float *bias1;         // [B, 256]: b1 + W1[:, :80] . (camera code, identity code)
    const __bf16 *w1pos;        // [256][48]: W1[:, 80:120], zero padded
    const __bf16 *wh;           // forward: [4][256][256] = W_l [out][in]; backward: W_l^T [in][out]
    const float *bh;            // [4][256]
    const float *w6;            // [3][256]
    const float *b6;            // [3]
    __bf16 *acts;               // [5][B*HW][256] post-activation outputs of the five hidden layers (NULL: not kept)
    float *out;                 // [B, 3, HW]
    const float *grad_out;      // [B, 3, HW]
    __bf16 *dz;                 // [5][B*HW][256] gradients w.r.t. the five pre-activations
    __bf16 *x0;                 // [B*HW][48] positional encoding as the first GEMM consumed it (kept for its weight gradient)
    float *colsum;              // [5][B*tiles_per_image][256] per-tile column sums of dz (bias gradients, summed by the caller)
};

// The weights of all layers form ONE stream of 16-channel chunks (forward: 3 chunks of W1's positional part, then 4 x 16
// of the hidden layers; backward: 4 x 16 of the transposed hidden layers, last layer first).  Chunk g of the stream is
// 8 KB: element (n, k) at n * 16 + k, thread t stages the 16 bytes (n = t >> 1, k = 8 (t & 1) ..).  `g` is a
// compile-time constant wherever this is called (fully unrolled loops), so the layer arithmetic folds away.
template <bool FWD>
__device__ __forceinline__ const __bf16 *chunk_ptr(const Params &p, int g, int tid) {
    if (FWD && g < kK0 / kChunk) return p.w1pos + (tid >> 1) * kK0 + (tid & 1) * 8 + g * kChunk;
    const int h = FWD ? g - kK0 / kChunk : g;
    const int l = FWD ? (h >> 4) : kHidden - 1 - (h >> 4), c = h & 15;
    return p.wh + (size_t)l * kWidth * kWidth + (tid >> 1) * kWidth + (tid & 1) * 8 + c * kChunk;
}
constexpr int kFwdChunks = kK0 / kChunk + kHidden * (kWidth / kChunk);  // 67
constexpr int kBwdChunks = kHidden * (kWidth / kChunk);                 // 64
constexpr int kQueue = 4;  // chunks in flight per thread (registers), i.e. an L2 round trip is covered by ~4 k-steps of MFMA

// acc[mi][ni] = (X[64 mq + 32 mi .., :K] . W[128 nh + 32 ni .., :K]^T)^T for the layer whose weights are chunks
// G0 .. G0 + K/16 - 1 of the stream.  The MFMA is issued with the WEIGHT fragment as A and the activation fragment as B,
// so a lane ends up with 4 consecutive output channels of one pixel per register quad (8-byte LDS writes in the
// epilogue instead of 2-byte ones).
// Software pipeline, one barrier per k-step: during step c a thread (1) writes its 16 bytes of chunk c + 2 from the
// register queue into ring buffer (c + 2) % 3 and re-fills that queue slot with chunk c + 6 from global memory, (2)
// issues the LDS reads of the fragments of step c + 1, (3) issues the 8 MFMAs of step c on fragments read one step
// earlier -- so neither the L2 round trip nor the LDS read latency sits between two MFMA groups.  The barrier that opens
// step c makes chunk c + 1 (written during step c - 1) visible; the buffer overwritten in step c held chunk c - 1, whose
// fragments every wave received before it issued its MFMAs of step c - 1, i.e. before that barrier.
// q[] is the register queue: on entry q[(G0 + j) % 4] holds chunk G0 + j (j < 4), on exit the same for the next layer
// (the prefetch runs across layer boundaries).  The caller guarantees the ring is idle on entry (a barrier since its
// last use, incl. its use as scratch).
// ST: the tile X that this GEMM READS is at the same time what HBM has to receive (forward: the previous layer's
// activations; backward: the gradient plane just masked; first forward layer: the positional encoding), so its rows leave
// here, one 16-byte unit per thread and k-step (ST = 1: [rows][256], 16 steps; ST = 2: [rows][48], 3 steps), AFTER that
// step's weight request.  Loads and stores retire through one in-order counter (vmcnt): a weight chunk requested after a
// burst of 16 stores per thread could only be waited for once all of them were acknowledged, which serialised every
// layer's store burst with the GEMM behind it (rounds 3-4: the matrix pipe idle 72 % / 86 % of the two kernels).  In this
// order the chunk staged in step c (requested in step c - 4) waits for the rows stored up to step c - 5 only.
template <int K, int G0, bool FWD, int ST>
__device__ __forceinline__ void tile_gemm(const __bf16 *X, __bf16 *Wr, const Params &p, bf16x8 (&q)[kQueue],
                                          f32x16 (&acc)[2][4], int mq, int nh, int lane, int tid,
                                          __bf16 *__restrict__ gdst = nullptr, int nvalid = 0) {
    static_assert(ST == 0 || (ST == 1 && K / kChunk == kMaskPasses) || (ST == 2 && K == kK0 && kTileM * (kK0 / 8) == 3 * kThreads),
                  "one store unit per thread and k-step");
    constexpr int NC = K / kChunk, GT = FWD ? kFwdChunks : kBwdChunks;
    const int lr = lane & 31, lk = (lane >> 5) * 8;
    const __bf16 *xa = X + (64 * mq + lr) * kLdX + lk;
    // ring buffer layout [k-half][n][8]: a wave's fragment read (32 consecutive n, one k-half per half-wave) is two
    // contiguous 512-byte blocks = conflict-free; with [n][16] the 16 lanes of a pass were 32 bytes apart (2-way conflicts)
    __bf16 *dst = Wr + ((tid & 1) * kWidth + (tid >> 1)) * 8;
    const __bf16 *wb = Wr + ((lane >> 5) * kWidth + 128 * nh + lr) * 8;
#define MVP_STAGE(G_)                                                                                         \
    {                                                                                                         \
        *reinterpret_cast<bf16x8 *>(dst + ((G_) % kRing) * kRingElems) = q[(G_) % kQueue];                    \
        if ((G_) + kQueue < GT)                                                                               \
            q[(G_) % kQueue] = *reinterpret_cast<const bf16x8 *>(chunk_ptr<FWD>(p, (G_) + kQueue, tid));      \
    }
#define MVP_FRAGS(A_, B_, C_)                                                                                 \
    {                                                                                                         \
        _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) B_[ni] =                                             \
            *reinterpret_cast<const bf16x8 *>(wb + ((G0 + (C_)) % kRing) * kRingElems + ni * 32 * 8);         \
        _Pragma("unroll") for (int mi = 0; mi < 2; ++mi) A_[mi] =                                             \
            *reinterpret_cast<const bf16x8 *>(xa + mi * 32 * kLdX + (C_) * kChunk);                           \
    }
    MVP_STAGE(G0)
    if (NC > 1) MVP_STAGE(G0 + 1)
    __syncthreads();
    bf16x8 a[2][2], b[2][4];  // [parity of the step][..]
    MVP_FRAGS(a[0], b[0], 0)
    const bool storing = ST != 0 && gdst != nullptr;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c > 0) __syncthreads();
        if (c + 2 < NC) MVP_STAGE(G0 + c + 2)
        if (c + 1 < NC) MVP_FRAGS(a[(c + 1) & 1], b[(c + 1) & 1], c + 1)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                if (c == 0) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[0][ni], a[0][mi], zero, 0, 0, 0);
                } else {
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[c & 1][ni], a[c & 1][mi], acc[mi][ni], 0, 0, 0);
                }
            }
        if (ST != 0) {  // (the row is read from LDS while the step's MFMAs execute.  Predicated on the row being inside
                        //  the tile; a straight-line version -- rows past the end clamped to the last valid one -- measured
                        //  3 % slower: it stretches the step's live ranges)
            const int si = tid + c * kThreads;
            const int srow = ST == 2 ? si / (kK0 / 8) : si >> 5, sc8 = ST == 2 ? (si - srow * (kK0 / 8)) * 8 : (si & 31) * 8;
            if (storing && srow < nvalid)
                *reinterpret_cast<bf16x8 *>(gdst + (unsigned)(srow * (ST == 2 ? kK0 : kWidth) + sc8)) =
                    *reinterpret_cast<const bf16x8 *>(X + srow * kLdX + sc8);
        }
    }
#undef MVP_STAGE
#undef MVP_FRAGS
}

// accumulators -> LDS tile as bf16; ACT: + bias[channel], LeakyReLU.  Register 4 u + e of tile (mi, ni) is channel
// 128 nh + 32 ni + 8 u + 4 (lane >> 5) + e of pixel 64 mq + 32 mi + (lane & 31): four consecutive channels per store.
template <bool ACT>
__device__ __forceinline__ void acc_to_lds(__bf16 *X, const f32x16 (&acc)[2][4], const float *__restrict__ bias, int mq,
                                           int nh, int lane) {
    // (opaque per call: otherwise the 16 bias addresses of the five epilogues are one set of loop invariants, kept -- and
    //  spilled -- across the whole tile)
    int nb = 128 * nh + 4 * (lane >> 5);
    asm volatile("; epilogue" : "+v"(nb));
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n0 = nb + 32 * ni + 8 * u;
            float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ACT) {
                bb = *reinterpret_cast<const float4 *>(bias + n0);
                asm volatile("" ::: "memory");  // keep the 16 bias loads of a layer from being hoisted (and spilled) together
            }
            const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int pix = 64 * mq + 32 * mi + (lane & 31);
                bf16x4 o;
#pragma unroll
                for (int h = 0; h < 2; ++h) {  // two channels per packed operation
                    f32x2 v = {acc[mi][ni][4 * u + 2 * h], acc[mi][ni][4 * u + 2 * h + 1]};
                    if (ACT) {
                        v += f32x2{bv[2 * h], bv[2 * h + 1]};
                        const f32x2 sl = v * f32x2{kSlope, kSlope};
                        v.x = fmaxf(v.x, sl.x), v.y = fmaxf(v.y, sl.y);  // LeakyReLU(0.2): max(v, 0.2 v)
                    }
                    o[2 * h] = (__bf16)v.x, o[2 * h + 1] = (__bf16)v.y;
                }
                *reinterpret_cast<bf16x4 *>(X + pix * kLdX + n0) = o;
            }
        }
    }
}

// LDS tile -> global [rows][256] bf16, 16 bytes per thread and pass (rows >= nvalid are not written)
__device__ __forceinline__ void lds_to_global(const __bf16 *X, __bf16 *__restrict__ G, int nvalid, int tid) {
#pragma unroll 4
    for (int i = tid; i < kTileM * (kWidth / 8); i += kThreads) {
        const int row = i >> 5, c8 = (i & 31) * 8;
        if (row < nvalid)
            *reinterpret_cast<bf16x8 *>(G + (size_t)row * kWidth + c8) = *reinterpret_cast<const bf16x8 *>(X + row * kLdX + c8);
    }
}

// TRAIN: the planes the backward needs (x0, acts) are written; false = inference, no store code at all
template <bool TRAIN>
__global__ __launch_bounds__(kThreads, 1) void fwd_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 *X = reinterpret_cast<__bf16 *>(smem);
    __bf16 *Wr = X + kTileM * kLdX;               // weight ring, 3 x 8 KB
    float *w6s = reinterpret_cast<float *>(Wr);   // [3][256] + [512][3], last layer only (the ring is idle then)
    float *red = w6s + 3 * kWidth;
    float *bl = reinterpret_cast<float *>(Wr + kRing * kRingElems);  // [256]: the bias of the layer being multiplied
    const int tid0 = threadIdx.x;
    // The first weight chunks and this thread's pixel of the tile are requested before anything else -- for the first tile
    // here, for later tiles at the end of the tile before, ahead of its last store burst.
    bf16x8 q[kQueue];
    float2 sc = make_float2(0.f, 0.f);
    if ((int)blockIdx.x < p.B * p.tiles_per_image) {
#pragma unroll
        for (int j = 0; j < kQueue; ++j) q[j] = *reinterpret_cast<const bf16x8 *>(chunk_ptr<true>(p, j, tid0));
        const int b0 = blockIdx.x / p.tiles_per_image, q0 = (blockIdx.x - b0 * p.tiles_per_image) * kTileM + (tid0 & (kTileM - 1));
        if (q0 < p.HW) sc = reinterpret_cast<const float2 *>(p.samplecoords)[(size_t)b0 * p.HW + q0];
    }
    // persistent: one workgroup per CU walks its share of the tiles (a launch per tile cost ~4 us of setup each)
    for (int tile = blockIdx.x; tile < p.B * p.tiles_per_image; tile += gridDim.x) {
    // (opaque per iteration: otherwise the ~70 per-thread weight-chunk addresses are hoisted out of the tile loop as
    //  loop invariants and spilled)
    int tid = tid0;
    asm volatile("; per-tile thread index" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int mh = wave & 3, nh = wave >> 2;
    const int b = tile / p.tiles_per_image, p0 = (tile - b * p.tiles_per_image) * kTileM;
    const int nvalid = min(kTileM, p.HW - p0);
    const size_t P = (size_t)p.B * p.HW, pix0 = (size_t)b * p.HW + p0;

    // ---- positional encoding of the tile (mlp2d.py:64-68): channel 2 i + j = sin(2^i pi x_j), 20 + 2 i + j = cos ----
    {
        const int r = tid & (kTileM - 1), half = tid / kTileM;  // half 0: sines, half 1: cosines + the zero padding
        const float x0 = r < nvalid ? sc.x : 0.f, x1 = r < nvalid ? sc.y : 0.f;
        __bf16 *row = X + r * kLdX + half * 20;
        // v_sin_f32 / v_cos_f32 take their argument in revolutions: sin(2^i pi x) = v_sin(2^(i-1) x).  Their valid
        // domain is |arg| <= 256 revolutions (outside it the hardware returns sin = 0, cos = 1), which 2^8 x leaves as
        // soon as |x| > 1 -- cropped, sub-sampled or jittered pixelcoords under the shape-based normalisation of
        // autoencoder.py:231-237 -- so the argument is reduced first: v_fract_f32 is exact and 2^(i-1) x is an exact
        // product, hence sin(2 pi fract(r)) == sin(2 pi r) for every finite x (torch.sin/cos have no such limit
        // either).  Absolute error ~1e-6, far below the bf16 rounding of the result.
        float f = 0.5f;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const float r0 = __builtin_amdgcn_fractf(f * x0), r1 = __builtin_amdgcn_fractf(f * x1);
            row[2 * i] = (__bf16)(half ? __builtin_amdgcn_cosf(r0) : __builtin_amdgcn_sinf(r0));
            row[2 * i + 1] = (__bf16)(half ? __builtin_amdgcn_cosf(r1) : __builtin_amdgcn_sinf(r1));
            f *= 2.f;
        }
        if (half) {  // channels 40..47 = 0 (16 bytes; the zero is made here: as a loop invariant it was kept in scratch)
            unsigned z = 0;
            asm volatile("" : "+v"(z));
            *reinterpret_cast<uint4 *>(X + r * kLdX + kPos) = make_uint4(z, z, z, z);
        }
    }
    __syncthreads();
    f32x16 acc[2][4];
    // ---- layer 1: 40 (48) -> 256, per-image bias; the positional encoding it reads goes to x0 (96 bytes per pixel) ----
    // (a layer's bias is requested BEFORE its GEMM and parked in LDS after it: 16 vector loads per thread in the epilogue
    //  would each wait for the rows the GEMM has just stored)
    float bv = p.bias1[(size_t)b * kWidth + (tid & (kWidth - 1))];
    tile_gemm<kK0, 0, true, TRAIN ? 2 : 0>(X, Wr, p, q, acc, mh, nh, lane, tid, p.x0 ? p.x0 + pix0 * kK0 : nullptr, nvalid);
    if (tid < kWidth) bl[tid] = bv;
    __syncthreads();
    acc_to_lds<true>(X, acc, bl, mh, nh, lane);
    __syncthreads();
    // ---- layers 2..5: 256 -> 256; each stores the activations it reads (the previous layer's output) ----
#define MVP_HIDDEN_LAYER(L_)                                                                              \
    {   /* (the plane's address is formed here, not kept in scalar registers from the top of the tile) */ \
        size_t row0 = (size_t)(L_) * P + pix0;                                                            \
        if (TRAIN) asm volatile("" : "+s"(row0));                                                         \
        bv = p.bh[(L_) * kWidth + (tid & (kWidth - 1))];                                                  \
        tile_gemm<kWidth, kK0 / kChunk + (L_) * (kWidth / kChunk), true, TRAIN ? 1 : 0>(                  \
            X, Wr, p, q, acc, mh, nh, lane, tid, p.acts ? p.acts + row0 * kWidth : nullptr, nvalid);      \
        if (tid < kWidth) bl[tid] = bv;                                                                   \
    }                                                                                                     \
    __syncthreads(); /* every wave has read the whole tile */                                             \
    if ((L_) == kHidden - 1) { /* the ring is idle: W6 for the last layer, and the next tile's first weight chunks, are  \
                                  requested before this tile's last store burst, not behind it */           \
        for (int i = tid; i < 3 * kWidth; i += kThreads) w6s[i] = p.w6[i];                                \
        const int tn = tile + (int)gridDim.x;                                                             \
        if (tn < p.B * p.tiles_per_image) {                                                               \
            _Pragma("unroll") for (int j = 0; j < kQueue; ++j)                                            \
                q[j] = *reinterpret_cast<const bf16x8 *>(chunk_ptr<true>(p, j, tid));                     \
            const int bn = tn / p.tiles_per_image, pn = (tn - bn * p.tiles_per_image) * kTileM;           \
            const int rn = tid & (kTileM - 1);                                                            \
            if (pn + rn < p.HW) sc = reinterpret_cast<const float2 *>(p.samplecoords)[(size_t)bn * p.HW + pn + rn]; \
        }                                                                                                 \
    }                                                                                                     \
    acc_to_lds<true>(X, acc, bl, mh, nh, lane);                                                           \
    __syncthreads();
    MVP_HIDDEN_LAYER(0)
    MVP_HIDDEN_LAYER(1)
    MVP_HIDDEN_LAYER(2)
    MVP_HIDDEN_LAYER(3)
#undef MVP_HIDDEN_LAYER
    if (TRAIN && p.acts) lds_to_global(X, p.acts + ((size_t)kHidden * P + pix0) * kWidth, nvalid, tid);
    // ---- layer 6: 256 -> 3, * 25 + 100 (mlp2d.py:40,70); W6 is in the ring already (per-thread global loads of it cost
    //      384 each) ----
    {
        const int r = tid & (kTileM - 1), half = tid / kTileM;
        const __bf16 *xr = X + r * kLdX + half * 128;
        const float *w = w6s + half * 128;  // same address in every lane of a wave: LDS broadcast reads
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 4
        for (int k = 0; k < 128; k += 8) {
            const bf16x8 xv = *reinterpret_cast<const bf16x8 *>(xr + k);
            float w0[8], w1[8], w2[8];
            *reinterpret_cast<float4 *>(w0) = *reinterpret_cast<const float4 *>(w + k);
            *reinterpret_cast<float4 *>(w0 + 4) = *reinterpret_cast<const float4 *>(w + k + 4);
            *reinterpret_cast<float4 *>(w1) = *reinterpret_cast<const float4 *>(w + kWidth + k);
            *reinterpret_cast<float4 *>(w1 + 4) = *reinterpret_cast<const float4 *>(w + kWidth + k + 4);
            *reinterpret_cast<float4 *>(w2) = *reinterpret_cast<const float4 *>(w + 2 * kWidth + k);
            *reinterpret_cast<float4 *>(w2 + 4) = *reinterpret_cast<const float4 *>(w + 2 * kWidth + k + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xf = (float)xv[e];
                s0 = fmaf(xf, w0[e], s0), s1 = fmaf(xf, w1[e], s1), s2 = fmaf(xf, w2[e], s2);
            }
        }
        red[tid * 3 + 0] = s0, red[tid * 3 + 1] = s1, red[tid * 3 + 2] = s2;
    }
    __syncthreads();
    if (tid < kTileM && tid < nvalid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = red[tid * 3 + c] + red[(tid + kTileM) * 3 + c] + p.b6[c];
            p.out[((size_t)b * 3 + c) * p.HW + p0 + tid] = v * 25.f + 100.f;
        }
    }
    __syncthreads();  // the next tile overwrites X and the ring scratch
    }
}

// Eight gradients (4 words of two bf16) times leaky'(activation), as bf16 again; `cs` += the eight products.  Per pair:
// two shifts / masks to fp32, the two slopes from the activations' bits (lower half: its fp32 value > 0; upper half: the
// word as a signed integer > 0xffff, i.e. sign clear and a non-zero bf16 -- a positive NaN counts as positive), one packed
// multiply, one packed add, one packed conversion: 5 operations per element less than element by element.
__device__ __forceinline__ uint4 mask8f(const f32x2 (&gv)[4], const uint4 a, f32x2 (&cs)[4]) {  // gradients given in fp32
    const unsigned aw[4] = {a.x, a.y, a.z, a.w};
    unsigned o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x2 m;
        m.x = __uint_as_float(aw[k] << 16) > 0.f ? 1.f : kSlope;
        m.y = (int)aw[k] > 0xffff ? 1.f : kSlope;
        const f32x2 pr = gv[k] * m;
        cs[k] += pr;
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        const bf16x2 r = {(__bf16)pr.x, (__bf16)pr.y};
        o[k] = __builtin_bit_cast(unsigned, r);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ uint4 mask8(const uint4 g, const uint4 a, f32x2 (&cs)[4]) {
    const unsigned gw[4] = {g.x, g.y, g.z, g.w};
    f32x2 gv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) gv[k].x = __uint_as_float(gw[k] << 16), gv[k].y = __uint_as_float(gw[k] & 0xffff0000u);
    return mask8f(gv, a, cs);
}

// dst tile (LDS, bf16) *= leaky'(A) with A from global; result also to global dz
// `cs` accumulates this thread's 8 columns (c8 = 8 (tid & 31) .. + 7) over its 16 rows: the bias gradient's partial sum
// STORE = false: the masked plane only goes back to LDS; the GEMM that reads it next stores it (tile_gemm, ST = 1)
template <bool STORE>
__device__ __forceinline__ void mask_and_store(__bf16 *X, const __bf16 *A, __bf16 *__restrict__ DZ, int nvalid, int tid,
                                               float (&cs)[8]) {
    // All of this thread's activation rows are requested BEFORE the first gradient row is stored: loads and stores
    // retire through one in-order counter (vmcnt), so a load issued after a store cannot be waited for without waiting
    // for that store's acknowledgement as well -- interleaved (2 rows per round), every round of this pass paid an HBM
    // read latency plus a write acknowledgement.  The 64 registers are free here (the accumulators are dead).
    // The two compiler fences keep the 16 requests where they are written: not above the barrier before this pass (the
    // accumulators still live there) and not sunk into the loop below, next to their uses (A is deliberately not
    // __restrict__, and rows past the tile's end are clamped instead of predicated, or the loads are free to move).
    // (and the 16 clamped offsets are recomputed in every pass -- `tid` made opaque -- or they are kept, and spilled, across
    //  the whole tile as invariants of the layer loop)
    asm volatile("; mask pass" : "+v"(tid) : : "memory");
    f32x2 cs2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cs2[k] = f32x2{cs[2 * k], cs[2 * k + 1]};
    uint4 av[kMaskPasses];
#pragma unroll
    for (int j = 0; j < kMaskPasses; ++j) {
        // (32-bit element offsets from the tile's uniform base)
        const unsigned i = tid + j * kThreads, off = min(i >> 5, (unsigned)nvalid - 1u) * kWidth + (i & 31u) * 8u;
        av[j] = *reinterpret_cast<const uint4 *>(A + off);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < kMaskPasses; ++j) {
        const int i = tid + j * kThreads, row = i >> 5, c8 = (i & 31) * 8;
        uint4 g = *reinterpret_cast<const uint4 *>(X + row * kLdX + c8);
        if (row < nvalid) {
            g = mask8(g, av[j], cs2);
            if (STORE) *reinterpret_cast<uint4 *>(DZ + (unsigned)(row * kWidth + c8)) = g;
        } else {
            g = make_uint4(0u, 0u, 0u, 0u);
        }
        *reinterpret_cast<uint4 *>(X + row * kLdX + c8) = g;
        asm volatile("" ::: "memory");  // one row at a time: 16 LDS rows read ahead next to the 16 requested ones would spill
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) cs[2 * k] = cs2[k].x, cs[2 * k + 1] = cs2[k].y;
}

// column sums of the tile: 16 threads (tid >> 5) hold partial sums of the same 8 columns -> LDS -> one row of `out`
__device__ __forceinline__ void reduce_colsum(float *scratch /*[16][256]*/, const float (&cs)[8], float *__restrict__ out, int tid) {
#pragma unroll
    for (int e = 0; e < 8; ++e) scratch[(tid >> 5) * kWidth + (tid & 31) * 8 + e] = cs[e];
    __syncthreads();
    if (tid < kWidth) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < kThreads / 32; ++j) s += scratch[j * kWidth + tid];
        out[tid] = s;
    }
    __syncthreads();
}

__global__ __launch_bounds__(kThreads, 1) void bwd_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 *X = reinterpret_cast<__bf16 *>(smem);
    __bf16 *Wr = X + kTileM * kLdX;                      // weight ring, 3 x 8 KB ...
    float *scratch = reinterpret_cast<float *>(Wr);      // ... doubling as the [16][256] column-sum staging between GEMMs
    float *gl = reinterpret_cast<float *>(Wr + kRing * kRingElems);  // [256][3] upstream gradient * 25
    const int tid0 = threadIdx.x;
    bf16x8 q[kQueue];  // (later tiles: requested at the end of the tile before, ahead of its last stores)
#pragma unroll
    for (int j = 0; j < kQueue; ++j) q[j] = *reinterpret_cast<const bf16x8 *>(chunk_ptr<false>(p, j, tid0));
    for (int tile = blockIdx.x; tile < p.B * p.tiles_per_image; tile += gridDim.x) {
    int tid = tid0;
    asm volatile("; per-tile thread index" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int mh = wave & 3, nh = wave >> 2;
    const int b = tile / p.tiles_per_image, p0 = (tile - b * p.tiles_per_image) * kTileM;
    const int nvalid = min(kTileM, p.HW - p0);
    const size_t P = (size_t)p.B * p.HW, pix0 = (size_t)b * p.HW + p0;

    if (tid < kTileM) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            gl[tid * 3 + c] = tid < nvalid ? 25.f * p.grad_out[((size_t)b * 3 + c) * p.HW + p0 + tid] : 0.f;
    }
    __syncthreads();
    const size_t ntiles = (size_t)p.B * p.tiles_per_image;
    float cs[8];
    // ---- dA5 = g . W6 (K = 3: VALU), dZ5 = dA5 * leaky'(A5) ----
    {
        const __bf16 *A = p.acts + (4 * P + pix0) * kWidth;  // (dZ5 itself is stored by the GEMM that reads it)
        f32x2 w6r[3][4];  // this thread's 8 columns of W6 (c8 below is the same in every pass), as pairs
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                w6r[c][k] = f32x2{p.w6[c * kWidth + (tid & 31) * 8 + 2 * k], p.w6[c * kWidth + (tid & 31) * 8 + 2 * k + 1]};
        f32x2 cs2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) cs2[k] = f32x2{0.f, 0.f};
        // (fence: W6 is requested before the activations, so waiting for it never means waiting for a later store)
        asm volatile("" ::: "memory");
        uint4 av[kMaskPasses];  // (requested before the first store, like mask_and_store)
#pragma unroll
        for (int j = 0; j < kMaskPasses; ++j) {
            const unsigned off = min((unsigned)(tid + j * kThreads) >> 5, (unsigned)nvalid - 1u) * kWidth + (tid & 31) * 8;
            av[j] = *reinterpret_cast<const uint4 *>(A + off);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < kMaskPasses; ++j) {
            const int row = (tid + j * kThreads) >> 5, c8 = (tid & 31) * 8;
            const float g0 = gl[row * 3], g1 = gl[row * 3 + 1], g2 = gl[row * 3 + 2];
            uint4 g = make_uint4(0u, 0u, 0u, 0u);
            if (row < nvalid) {
                f32x2 da[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) da[k] = f32x2{g0, g0} * w6r[0][k] + f32x2{g1, g1} * w6r[1][k] + f32x2{g2, g2} * w6r[2][k];
                g = mask8f(da, av[j], cs2);
            }
            *reinterpret_cast<uint4 *>(X + row * kLdX + c8) = g;
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) cs[2 * k] = cs2[k].x, cs[2 * k + 1] = cs2[k].y;
    }
    __syncthreads();
    reduce_colsum(scratch, cs, p.colsum + (4 * ntiles + tile) * kWidth, tid);
    // ---- dZ_l = (dZ_{l+1} . W_{l+1}) * leaky'(A_l), l = 4..1 ----
    f32x16 acc[2][4];
#define MVP_BWD_LAYER(I_)                                                                                            \
    {                                                                                                                \
        constexpr int l = kHidden - 1 - (I_);                                                                        \
        /* transposed W; the plane dZ_(l+1) it reads goes to HBM from here */                                        \
        tile_gemm<kWidth, (I_) * (kWidth / kChunk), false, 1>(X, Wr, p, q, acc, mh, nh, lane, tid,                   \
                                                              p.dz + ((size_t)(l + 1) * P + pix0) * kWidth, nvalid); \
        __syncthreads();                                                                                             \
        acc_to_lds<false>(X, acc, nullptr, mh, nh, lane);                                                            \
        __syncthreads();                                                                                             \
        for (int e = 0; e < 8; ++e) cs[e] = 0.f;                                                                     \
        if (l == 0 && tile + (int)gridDim.x < p.B * p.tiles_per_image) { /* next tile's weights: ahead of the stores */ \
            _Pragma("unroll") for (int j = 0; j < kQueue; ++j)                                                       \
                q[j] = *reinterpret_cast<const bf16x8 *>(chunk_ptr<false>(p, j, tid));                               \
        }                                                                                                            \
        mask_and_store<l == 0>(X, p.acts + ((size_t)l * P + pix0) * kWidth, p.dz + ((size_t)l * P + pix0) * kWidth,  \
                               nvalid, tid, cs);                                                                     \
        __syncthreads();                                                                                             \
        reduce_colsum(scratch, cs, p.colsum + ((size_t)l * ntiles + tile) * kWidth, tid);                           \
    }
    MVP_BWD_LAYER(0)
    MVP_BWD_LAYER(1)
    MVP_BWD_LAYER(2)
    MVP_BWD_LAYER(3)
#undef MVP_BWD_LAYER
    }  // (reduce_colsum ends with a barrier: X, ring and gl are free for the next tile)
}

}  // namespace bgmlp
}  // namespace mvp

// one workgroup per CU (its 156 KB of LDS fills the CU), each walking tiles blockIdx.x, blockIdx.x + grid, ...
static int persistent_grid(int ntiles) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus <= 0) {
        (void)hipGetLastError();
        cus = 256;
    }
    return ntiles < cus ? ntiles : cus;
}

static int bgmlp_common(int B, int HW, mvp::bgmlp::Params &p) {
    if (B < 0 || HW < 0) return MVP_ERR_BADARG;
    if ((long long)B * HW == 0) return 1;
    p.B = B, p.HW = HW;
    p.tiles_per_image = (HW + mvp::bgmlp::kTileM - 1) / mvp::bgmlp::kTileM;
    if ((long long)B * p.tiles_per_image > 0x7fffffffll) return MVP_ERR_UNSUPPORTED;
    return MVP_OK;
}

extern "C" int mvp_bgmlp_forward(int B, int HW, const float *samplecoords, const float *bias1, const void *w1pos,
                                 const void *wh, const float *bh, const float *w6, const float *b6, void *acts,
                                 void *x0, float *out, void *stream) {
    using namespace mvp;
    using namespace mvp::bgmlp;
    Params p = {};
    int rc = bgmlp_common(B, HW, p);
    if (rc == 1) return MVP_OK;
    if (rc != MVP_OK) return rc;
    if (!samplecoords || !bias1 || !w1pos || !wh || !bh || !w6 || !b6 || !out) return MVP_ERR_BADARG;
    if (!aligned16(w1pos) || !aligned16(wh) || (acts && !aligned16(acts)) || (x0 && !aligned16(x0)) ||
        !aligned16(bias1) || !aligned16(bh) || ((uintptr_t)samplecoords & 7u))
        return MVP_ERR_BADARG;
    if ((acts != nullptr) != (x0 != nullptr)) return MVP_ERR_BADARG;  // the two training planes go together
    p.samplecoords = samplecoords, p.bias1 = bias1, p.w1pos = static_cast<const __bf16 *>(w1pos);
    p.wh = static_cast<const __bf16 *>(wh), p.bh = bh, p.w6 = w6, p.b6 = b6;
    p.acts = static_cast<__bf16 *>(acts), p.x0 = static_cast<__bf16 *>(x0), p.out = out;
    const size_t lds = (size_t)kTileM * kLdX * 2 + kRing * kRingElems * 2 + kWidth * sizeof(float);
    // 157 KB of dynamic LDS: above the 64 KB default limit (gfx950 has 160 KB per CU)
    const void *fn = acts ? reinterpret_cast<const void *>(fwd_kernel<true>) : reinterpret_cast<const void *>(fwd_kernel<false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const dim3 grid((unsigned)persistent_grid(B * p.tiles_per_image));
    if (acts)
        hipLaunchKernelGGL(fwd_kernel<true>, grid, dim3(kThreads), lds, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(fwd_kernel<false>, grid, dim3(kThreads), lds, (hipStream_t)stream, p);
    return launch_status();
}

extern "C" int mvp_bgmlp_backward(int B, int HW, const float *grad_out, const void *acts, const void *whT, const float *w6,
                                  void *dz, float *colsum, void *stream) {
    using namespace mvp;
    using namespace mvp::bgmlp;
    Params p = {};
    int rc = bgmlp_common(B, HW, p);
    if (rc == 1) return MVP_OK;
    if (rc != MVP_OK) return rc;
    if (!grad_out || !acts || !whT || !w6 || !dz || !colsum) return MVP_ERR_BADARG;
    if (!aligned16(acts) || !aligned16(whT) || !aligned16(dz)) return MVP_ERR_BADARG;
    p.grad_out = grad_out, p.acts = const_cast<__bf16 *>(static_cast<const __bf16 *>(acts));
    p.wh = static_cast<const __bf16 *>(whT), p.w6 = w6, p.dz = static_cast<__bf16 *>(dz), p.colsum = colsum;
    const size_t lds = (size_t)kTileM * kLdX * 2 + kRing * kRingElems * 2 + kTileM * 3 * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(bwd_kernel, dim3((unsigned)persistent_grid(B * p.tiles_per_image)), dim3(kThreads), lds, (hipStream_t)stream, p);
    return launch_status();
}
