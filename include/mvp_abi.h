/*
 * include/mvp_abi.h -- C ABI of libmvp_gfx950.so, the MI355X-native (gfx950 / CDNA4) replacement of
 * the reference's two CUDA extensions on the MVP-raymarch training path.
 *
 * Drop-in boundary.  The reference binds its kernels through two pybind11 modules:
 *   mvpraymarchlib {compute_aabb, raymarch_forward, raymarch_backward, (compute_morton, build_tree: dead)}
 *       /root/reference/extensions/mvpraymarch/mvpraymarch.cpp:398-405
 *   utilslib       {compute_raydirs_forward, (compute_raydirs_backward: writes nothing)}
 *       /root/reference/extensions/utils/utils.cpp:134-137
 * whose bodies unwrap torch::Tensor into raw float and int pointers and call the launchers declared at
 *   mvpraymarch.cpp:12-100 (compute_aabb_cuda, raymarch_forward_cuda, raymarch_backward_cuda) and
 *   utils.cpp:12-40      (compute_raydirs_forward_cuda).
 * The entry points below ARE those launchers, re-cut for HIP: plain pointers + sizes, an explicit
 * hipStream_t (the reference launches on legacy stream 0: mvpraymarch.cpp:121,141,175,277,393), an int
 * status instead of silent failure, and no allocation inside the library (the reference's
 * compute_aabb_cuda does cudaMalloc/cudaMemset/cudaFree per call: bvh.cu:261-263,293).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to a dense, contiguous float32 (or uint32) array;
 *   - the library allocates nothing, keeps no global state and is re-entrant;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream) and the
 *     call returns without synchronising;
 *   - return value: MVP_OK (0) or a negative MVP_ERR_* code; positive values are hipError_t codes of a
 *     failed launch.  mvp_error_string() names either kind.
 *
 * Layouts (identical to the tensors the reference's Python hands to its bindings)
 *   raypos, raydir  [N,H,W,3]   tminmax [N,H,W,2]   rayrgba / grad_rayrgba [N,H,W,4]   raysat [N,H,W,3]
 *   primpos [N,K,3]  primrot [N,K,3,3] (row-major, rows R0,R1,R2)  primscale [N,K,3] (inverse half-extents)
 *   tplate / grad_tplate [N,K,TD,TH,TW,4]  (channels-last RGBA slabs, mvpraymarch.py:120-124)
 *   nodeaabb [N,2K-1,2,3]: implicit heap, internal i -> children 2i+1, 2i+2, leaf node K-1+k = primitive k
 *                          (the "fixedorder" tree, mvpraymarch.py:44-45,57-75,81)
 */
#ifndef MVP_ABI_H_
#define MVP_ABI_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVP_ABI_VERSION 17

#define MVP_OK 0
#define MVP_ERR_BADARG (-1)      /* null pointer / non-positive size / non-finite scalar            */
#define MVP_ERR_UNSUPPORTED (-2) /* a shape this build does not implement (e.g. slab dimension < 2) */
#define MVP_ERR_NODEVICE (-3)    /* no usable HIP device                                             */

/* number of uint32 words behind the optional `diag` pointer of the march entry points */
#define MVP_DIAG_WORDS 8
#define MVP_DIAG_FRONTIER_OVERFLOW 0 /* ray packets whose BFS frontier exceeded 512 -> exact DFS traversal      */
#define MVP_DIAG_LIST_OVERFLOW 1     /* ray packets whose hit list exceeded 512 (reference cap, utils.h:779) */
#define MVP_DIAG_SLOWPATH_PACKETS 2  /* hit packets marched by the slot-synchronous sweep (over a fast-path limit) */
#define MVP_DIAG_MAX_LIST 3          /* max hit-list length over all packets                             */
#define MVP_DIAG_PACKETS_HIT 4       /* packets with a non-empty hit list                                */
#define MVP_DIAG_LIST_ENTRIES 5      /* sum of hit-list lengths over all packets                         */
#define MVP_DIAG_CANDIDATES 6        /* sum of BVH candidate counts over all packets (before the exact test) */

int mvp_abi_version(void);
const char *mvp_error_string(int code);
/* gfx arch name of device `device` into buf (e.g. "gfx950"); MVP_ERR_NODEVICE when there is none. */
int mvp_device_arch(int device, char *buf, int buflen);

/* Ray generation.  Replaces compute_raydirs_forward_cuda (utils.cpp:12-24, utils_kernel.cu:12-52,97-129).
 * pixelcoords may be NULL: pixel (w,h) is used (utils_kernel.cu:36).  There is no backward: the
 * reference's backward kernel writes nothing (utils_kernel.cu:54-95; extensions/utils/utils.py:44-46). */
int mvp_raydirs_forward(int N, int H, int W, const float *campos /*[N,3]*/, const float *camrot /*[N,3,3]*/,
                        const float *focal /*[N,2]*/, const float *princpt /*[N,2]*/,
                        const float *pixelcoords /*[N,H,W,2] or NULL*/, float volradius, float *raypos,
                        float *raydir, float *tminmax, void *stream);

/* AABBs of the implicit heap BVH.  Replaces compute_aabb_cuda (mvpraymarch.cpp:26-36, bvh.cu:157-201,250-294)
 * for the "fixedorder" topology, which needs no sortedobjid / nodechildren / nodeparent tensors. */
int mvp_aabb_build(int N, int K, const float *primpos, const float *primrot, const float *primscale,
                   float *nodeaabb /*[N,2K-1,2,3]*/, void *stream);

/* Forward march.  Replaces raymarch_forward_cuda (mvpraymarch.cpp:38-66, mvpraymarch_kernel.cu:35-120,
 * mvpraymarch_subset_kernel.h:7-100) for algo 0 / fixedorder / channels-last / additive accumulation,
 * the instantiation the training path reaches, and -- when `warp` is given -- for algo 1, the warp-field
 * sampler PrimSamplerTW<true> (primsampler.h:53-58; mvpraymarch_kernel.cu:83-88,102): warp is [N,K,WD,WH,WW,3]
 * channels-last, the slab is sampled at warp(y) while the fade uses y.  raysat may be NULL (no-grad mode,
 * mvpraymarch.py:147-152); when given it is fully written (-1 where the ray never saturates).
 *
 * Grad-mode hand-off to the backward (all three may be NULL; then the backward uses its ray-centric path):
 *   rayaux          [N,H,W,4] uint32: {key of the saturating sample or 0xffffffff, bits(alpha before it), first lattice
 *                   step, bits(rtmax + 1e-5)} -- written for the rays of every 8x8 packet that lists a primitive (the only
 *                   ones the backward reads, through the list records that name them); other entries are left as they were
 *   primlist_count  [N*K + 3 + N*ceil(H/8)*ceil(W/8)] uint32; the first N*K + 3 words are zeroed HERE (on `stream`) then
 *                   filled: packets per primitive; a flags word; a reserved word; bits(max |raysat|).  The rest is
 *                   scratch of the BACKWARD (per 8x8 ray packet: bits(max |grad_rayrgba|), rewritten by every call).
 *                   The backward may be called several times over one forward (retain_graph): what it marks in
 *                   this buffer (counter bits 30-31, flag bits 2-3) it clears again at the start of the next call.
 *   primlist        [N*K, primlist_cap, 4] uint32 (ABI 17; 2 words per record before): per primitive one 16-byte record per ray packet
 *                   that touches it -- {(packet << 9) | list slot, first | last << 16 lattice step of the packet in the box,
 *                   low and high word of the mask of the packet's 64 rays that have a step there}; 16-byte aligned;
 *                   primlist_cap must be a multiple of 4
 * diag may be NULL; otherwise MVP_DIAG_WORDS uint32 counters are ACCUMULATED into it. */
int mvp_march_forward(int N, int H, int W, int K, const float *raypos, const float *raydir, float stepsize,
                      const float *tminmax, const float *nodeaabb, const float *primpos, const float *primrot,
                      const float *primscale, int TD, int TH, int TW, const float *tplate, int WD, int WH, int WW,
                      const float *warp /*or NULL*/, float *rayrgba, float *raysat, uint32_t *rayaux,
                      uint32_t *primlist_count, uint32_t *primlist, int primlist_cap, float fadescale, float fadeexp,
                      uint32_t *diag, void *stream);

/* Forward march with the rays made inside the kernel (optional -- the drop-in path is
 * the two calls above).  Fuses compute_raydirs_forward_cuda (utils_kernel.cu:12-52) into raymarch_forward_cuda for the
 * caller models/autoencoder.py:240-252: no raypos / raydir / tminmax tensors are written or read (32 B per ray each
 * way).  The rays are bit-identical to mvp_raydirs_forward's (one shared statement of the arithmetic), hence so is
 * rayrgba.  pixelcoords may be NULL (integer pixel grid).  algo 0 only (no warp field).
 * Training: mvp_march_backward takes ray tensors.  With raypos_out / raydir_out / tminmax_out ([N,H,W,3], [N,H,W,3],
 * [N,H,W,2]; all three or all NULL) the forward writes the rays it made -- every pixel by the one packet that owns it,
 * the bytes mvp_raydirs_forward would have written -- and the caller hands them to mvp_march_backward: the training
 * step then has no raydirs launch and the forward reads no ray tensor. */
int mvp_march_forward_cams(int N, int H, int W, int K, const float *campos /*[N,3]*/, const float *camrot /*[N,3,3]*/,
                           const float *focal /*[N,2]*/, const float *princpt /*[N,2]*/,
                           const float *pixelcoords /*[N,H,W,2] or NULL*/, float volradius, float stepsize,
                           const float *nodeaabb, const float *primpos, const float *primrot, const float *primscale,
                           int TD, int TH, int TW, const float *tplate, float *rayrgba, float *raysat, uint32_t *rayaux,
                           uint32_t *primlist_count, uint32_t *primlist, int primlist_cap,
                           float *raypos_out /*or NULL*/, float *raydir_out /*or NULL*/, float *tminmax_out /*or NULL*/,
                           float fadescale, float fadeexp, uint32_t *diag, void *stream);

/* Opt-in render path over HALF-PRECISION slabs (round 5; no counterpart in the reference, whose kernels read fp32 slabs:
 * primsampler.h:44-66, utils.h:408-502).  tplate_half is [N,K,8,8,8,4] fp16 RGBA (8 bytes per voxel), made by
 * mvp_template_to_half or mvp_template_assemble_forward_half below.  Same sample set, weights, interpolation and
 * compositing as mvp_march_forward in fp32 arithmetic; the only difference is the storage rounding of the slab values
 * (2^-11 relative).  The forward's sweep is bound by the texture path per lane-gather: the fp16 layout makes the two x
 * neighbours of a corner pair one 16-byte gather -- 4 gathers per sample instead of 8, over half the bytes.
 * Rays: either tensors (raypos, raydir, tminmax; campos == NULL) or made inside the march (campos .. princpt, optional
 * pixelcoords, volradius; raypos == raydir == tminmax == NULL).  Forward only (no raysat, no hand-off to a backward);
 * 8^3 slabs, else MVP_ERR_UNSUPPORTED. */
int mvp_march_render_half(int N, int H, int W, int K, const float *raypos, const float *raydir, const float *tminmax,
                          const float *campos, const float *camrot, const float *focal, const float *princpt,
                          const float *pixelcoords, float volradius, float stepsize, const float *nodeaabb,
                          const float *primpos, const float *primrot, const float *primscale, int TD, int TH, int TW,
                          const void *tplate_half, float *rayrgba, float fadescale, float fadeexp, uint32_t *diag,
                          void *stream);
/* fp32 slab tensor -> fp16 (round to nearest even), `voxels` = N*K*TD*TH*TW (even).  16 B read + 8 B written per voxel. */
int mvp_template_to_half(long long voxels, const float *tplate, void *tplate_half, void *stream);

/* Backward march.  Replaces raymarch_backward_cuda (mvpraymarch.cpp:68-100, mvpraymarch_kernel.cu:122-207,
 * mvpraymarch_subset_kernel.h:102-216).  The four grad buffers are OVERWRITTEN (every element is written; the
 * caller need not zero-fill them, unlike mvpraymarch.py:240-246).  With the forward's hand-off buffers the
 * primitive-centric kernel runs (no HBM atomics; with a warp field -- algo 1 -- its warp-field variant, and grad_warp is
 * overwritten too); primitives whose list overflowed primlist_cap, or everything when the buffers are NULL / the
 * slab(s) exceed the LDS budget, go through the ray-centric kernel with global_atomic_add_f32. */
int mvp_march_backward(int N, int H, int W, int K, const float *raypos, const float *raydir, float stepsize,
                       const float *tminmax, const float *nodeaabb, const float *primpos, const float *primrot,
                       const float *primscale, int TD, int TH, int TW, const float *tplate, int WD, int WH, int WW,
                       const float *warp /*or NULL*/, const float *raysat, const uint32_t *rayaux,
                       uint32_t *primlist_count, const uint32_t *primlist, int primlist_cap,
                       const float *grad_rayrgba, float *grad_primpos, float *grad_primrot, float *grad_primscale,
                       float *grad_tplate, float *grad_warp /*NULL iff warp is NULL*/, float fadescale,
                       float fadeexp, uint32_t *diag, void *stream);

/* The decode tail and its image loss in one pass each way: colour calibration `w * image + b`
 * (models/colorcals/colorcal.py:28-31, models/autoencoder.py:254-256), matting `rayrgb + (1 - rayalpha) * bg`
 * (autoencoder.py:263-265) and sum |irgbrec - image| (losses.py:12-14, ddp-train.py:404-405), read from / written to the
 * march's own layout rayrgba [N,H,W,4].  cw, cb [N,3] (both or neither), bg, target [N,3,H,W] may be NULL.  irgbrec
 * [N,3,H,W] is bit-identical to the eager statements.  l1_partials [N * mvp_pixel_tail_blocks(H, W)] (NULL without a
 * target) receives one partial sum per workgroup; cwcb_partials [N, blocks, 6] the partial sums of the gradients of
 * (cw, cb).  g_l1: DEVICE scalar, the upstream gradient of the L1 sum (NULL = 0); g_irgbrec / g_ialpha may be NULL. */
int mvp_pixel_tail_blocks(int H, int W);
int mvp_pixel_tail_forward(int N, int H, int W, const float *rayrgba, const float *cw, const float *cb, const float *bg,
                           const float *target, float *irgbrec, float *ialpha, float *l1_partials, void *stream);
int mvp_pixel_tail_backward(int N, int H, int W, const float *rayrgba, const float *cw, const float *bg, const float *target,
                            const float *irgbrec, const float *g_irgbrec, const float *g_ialpha, const float *g_l1,
                            float *grad_rayrgba, float *grad_bg, float *cwcb_partials, void *stream);

/* Demand statistics of the forward -> backward packet lists (no counterpart in the reference: its backward re-marches
 * every ray, mvpraymarch_subset_kernel.h:102-216).  `primlist_count` [nprims] as the grad-mode forward left it (the
 * counters keep counting past primlist_cap).  Writes hist[0..255] = number of primitives whose count, clamped to 2047,
 * lies in [8 b, 8 b + 8), hist[256] = the largest count.  `hist` (257 words) is zeroed by the call. */
int mvp_list_demand(const uint32_t *primlist_count, long long nprims, uint32_t *hist /*[257]*/, void *stream);

/* Which thread block of the march grids does what -- a host-side evaluation of the same functions the kernels use
 * (tests and tools; no device work).  The grids are XCD-aware: block b runs on XCD b % 8, whole images go to single XCDs
 * and the rest are shared.  kind 0: forward / ray-centric grid, out[i] = {image, 8x8 packet index};
 * kind 1: primitive-centric backward grid, out[i] = {image, primitive}; {-1, -1} for a block with nothing to do.
 * `count` entries from block `first_block` on are written; *total_blocks (may be NULL) receives the grid size. */
int mvp_march_block_map(int N, int H, int W, int K, int kind, int first_block, int count, int *out /*[count][2]*/,
                        int *total_blocks);

/* Decoder -> raymarch hand-off (no counterpart in the reference's extensions: there it is eager PyTorch spread over
 * models/decoders/rgb.py:137-143, models/decoders/geometry.py:183-185 and models/decoders/assembler.py:261).
 *   tex     [N, 3*B, nh*B, nh*B]  RGB decoder output (conv output + bias), channel index = z*3 + c
 *   opacity [N,   B, nh*B, nh*B]  geometry decoder opacity, channel index = z
 *   tplate  [N, nh*nh, B, B, B, 4] = cat(relu(rgb*25+100), relu(opacity)) in the march's channels-last slab layout
 * One pass, bit-identical to the eager expression.  Backward: grad_tex / grad_opacity are fully written. */
int mvp_template_assemble_forward(int N, int nh, int B, const float *tex, const float *opacity, float *tplate,
                                  void *stream);
int mvp_template_assemble_backward(int N, int nh, int B, const float *tplate, const float *grad_tplate,
                                   float *grad_tex, float *grad_opacity, void *stream);
/* The same forward writing fp16 RGBA slabs [N, nh*nh, B, B, B, 4] for mvp_march_render_half: the values of
 * mvp_template_assemble_forward rounded to nearest even, 8 bytes written per voxel instead of 16. */
int mvp_template_assemble_forward_half(int N, int nh, int B, const float *tex, const float *opacity, void *tplate_half,
                                       void *stream);
/* Frame-broadcast form of the same hand-off: ONE decoder output (tex [1,3*B,S,S], opacity [1,B,S,S]) shared by F frames,
 * frame f scaled by gain[f] in all four channels:  tplate [F, nh*nh, B,B,B, 4] = gain[f] * assemble(tex, opacity).
 * (User: the stand-in decoder of the train leg, ava-256_amd/trainloop.py; the reference's decoders emit per-frame outputs
 * and use the form above.)  Backward reads grad_tplate once: grad_tex / grad_opacity fully written (gradient of the shared
 * output = sum_f gain[f] * grad_tplate[f] through the relu), gain_partials [blocks, F] = per-workgroup partial sums of
 * <grad_tplate[f], base> -- the caller sums over blocks (blocks = mvp_template_assemble_frames_blocks(nh, B)); F <= 1024. */
long long mvp_template_assemble_frames_blocks(int nh, int B);
int mvp_template_assemble_frames_forward(int F, int nh, int B, const float *tex, const float *opacity, const float *gain,
                                         float *tplate, void *stream);
int mvp_template_assemble_frames_backward(int F, int nh, int B, const float *tex, const float *opacity, const float *gain,
                                          const float *grad_tplate, float *grad_tex, float *grad_opacity,
                                          float *gain_partials, void *stream);

/* Residual half of the hand-off (models/decoders/assembler.py:241-253, eager in the reference; `rodrig` = models/utils.py
 * Rodrigues): rw = clamp(residuals_weight, 0, 1);  if rw < 1: posres *= rw, rotres *= rw, scaleres = scaleres*rw + (1-rw);
 *   primpos = pos0 + posres;   primrot = pos-wise bmm(rot0, rodrig(rotres));   primscale = scale0 * scaleres.
 * pos0 / posres / rotres (axis-angle) / scaleres: [N or 1, K, 3]; rot0: [N or 1, K, 3, 3]; *_sn = frame stride in floats
 * (K*3 resp. K*9, or 0 = one array shared by the N frames); scale0: any broadcast of [N, K, 3] given by its three strides
 * (a scalar: 0,0,0; adaptwarps*0.8 [K]: 0,1,0).  Outputs [N,K,3], [N,K,3,3], [N,K,3], fully written; rw in [0, 1].
 * Backward: gradients shaped like their inputs, fully written (a shared input's gradient is the sum over the frames);
 * grad_pos0 / grad_rot0 may be NULL (not needed); scale0 gets none (a buffer in the reference: assembler.py:66,183-199). */
int mvp_prim_residuals_forward(int N, int K, float rw, const float *pos0, long long pos0_sn, const float *rot0,
                               long long rot0_sn, const float *scale0, long long scale0_sn, long long scale0_sk,
                               long long scale0_sc, const float *posres, long long posres_sn, const float *rotres,
                               long long rotres_sn, const float *scaleres, long long scaleres_sn, float *primpos,
                               float *primrot, float *primscale, void *stream);
int mvp_prim_residuals_backward(int N, int K, float rw, const float *pos0, long long pos0_sn, const float *rot0,
                                long long rot0_sn, const float *scale0, long long scale0_sn, long long scale0_sk,
                                long long scale0_sc, const float *posres, long long posres_sn, const float *rotres,
                                long long rotres_sn, const float *scaleres, long long scaleres_sn,
                                const float *grad_primpos, const float *grad_primrot, const float *grad_primscale,
                                float *grad_pos0 /*or NULL*/, float *grad_rot0 /*or NULL*/, float *grad_posres,
                                float *grad_rotres, float *grad_scaleres, void *stream);

/* The TBN frame of the hand-off (models/decoders/assembler.py:226-239, eager in the reference): from the centre texel's
 * differences vcenterdu / vcenterdv [M, 3] (M = frames x primitives; what mvp_prim_placement_forward returns) the base
 * orientation primrot [M, 3, 3] whose COLUMNS are tangent = du/|du|, bitangent = unit(normal x tangent), normal =
 * unit(tangent x dv), every norm clamped at 1e-8 like the reference's.  Backward: grad_du / grad_dv fully written. */
int mvp_prim_frame_forward(long long M, const float *du, const float *dv, float *primrot, void *stream);
int mvp_prim_frame_backward(long long M, const float *du, const float *dv, const float *grad_primrot, float *grad_du,
                            float *grad_dv, void *stream);

/* NHWC -> NCHW split of the march result.  Replaces `rayrgba.permute(0,3,1,2)` + `[:, :3].contiguous()` +
 * `[:, 3:4].contiguous()` of /root/reference/models/raymarchers/mvpraymarcher.py:50-51 (and autograd's slice / copy
 * backward) by one pass each way.  rayrgba [N,H,W,4] -> rayrgb [N,3,H,W], rayalpha [N,1,H,W]; the backward writes
 * grad_rayrgba [N,H,W,4] from grad_rayrgb / grad_rayalpha (either may be NULL = zero).  Bit-exact (data movement). */
int mvp_rgba_split_forward(int N, int H, int W, const float *rayrgba, float *rayrgb, float *rayalpha, void *stream);
int mvp_rgba_split_backward(int N, int H, int W, const float *grad_rayrgb /*or NULL*/,
                            const float *grad_rayalpha /*or NULL*/, float *grad_rayrgba, void *stream);

/* Primitive placement on the mesh (the barycentric half of SURVEY.md 8f row N2).  Replaces the eager expression of
 * /root/reference/models/decoders/assembler.py:118-122 -- a 1024 x 1024 x 3 position map per batch element from three
 * index_selects of [B, 1048576, 3] -- together with the only reads the assembler makes of that map
 * (assembler.py:143-206): the texel at each primitive's centre and its +u / +v neighbours.
 *   geo [B,V,3] float32 (already de-normalised: geo * vertstd + vertmean, assembler.py:101)
 *   idxim [T,T,3] int32 vertex indices, barim [T,T,3] float32 barycentric weights (assembler.py:61-64)
 *   centre of primitive k = i*nx + j is the texel (y0 + i*sy, x0 + j*sx); e.g. 16384 primitives: ny = nx = 128,
 *   y0 = x0 = 4, sy = sx = 8 (assembler.py:180); 256: 16 x 16, 32, 64 (:143)
 *   primpos [B,K,3] = postex(c);  vcenterdu [B,K,3] = postex(c+(0,1)) - postex(c);  vcenterdv = postex(c+(1,0)) - postex(c)
 * Forward is bit-identical to the eager expression.  Backward OVERWRITES grad_geo [B,V,3] (fp32 atomics, like the
 * reference's index_add); any of the three incoming gradients may be NULL (= zero). */
int mvp_prim_placement_forward(int B, int V, int T, int ny, int nx, int y0, int sy, int x0, int sx, float volradius,
                               const float *geo, const int *idxim, const float *barim, float *primpos,
                               float *vcenterdu, float *vcenterdv, void *stream);
int mvp_prim_placement_backward(int B, int V, int T, int ny, int nx, int y0, int sy, int x0, int sx, float volradius,
                                const int *idxim, const float *barim, const float *grad_primpos,
                                const float *grad_vcenterdu, const float *grad_vcenterdv, float *grad_geo,
                                void *stream);

/* Gradient hygiene of the optimisation loop as two multi-tensor passes.  Replaces, per
 * iteration, the per-parameter eager sequence of /root/reference/ddp-train.py:434-441:
 *     p.grad.data[torch.isnan(p.grad.data)] = 0 ; p.grad.data[torch.isinf(p.grad.data)] = 0   (for every parameter)
 *     torch.nn.utils.clip_grad_norm_(model.parameters(), clip)      (PyTorch: 2-norm of the per-tensor 2-norms,
 *                                                                     coef = min(1, clip / (norm + 1e-6)))
 * `grads` and `numels` are HOST arrays of `ntensors` device pointers (float32, 4-byte aligned, dense) and element
 * counts; they are read during the call only.  `sqnorm` is one double in DEVICE memory owned by the caller.
 *   mvp_grads_sanitize_sqnorm : non-finite elements become 0 in place; *sqnorm = sum of squares of the result.
 *   mvp_grads_clip_scale      : coef from *sqnorm on the device (no host sync); grads *= coef when coef < 1;
 *                               *total_norm (device float, may be NULL) = sqrt(*sqnorm). */
int mvp_grads_sanitize_sqnorm(int ntensors, float *const *grads, const long long *numels, double *sqnorm,
                              void *stream);
int mvp_grads_clip_scale(int ntensors, float *const *grads, const long long *numels, const double *sqnorm,
                         float max_norm, float *total_norm /*or NULL*/, void *stream);

/* Per-pixel background MLP of the training loop (second half) as two fused MFMA kernels.  No
 * native counterpart in the reference: there it is `BackgroundModelSimple.mlp`, six 1x1 Conv2d with LeakyReLU(0.2)
 * (/root/reference/models/bg/mlp2d.py:29-41), applied to cat(camera code, identity code, positional encoding) at
 * :61-70, output * 25 + 100.  The two 40-channel codes are constant over an image: they enter as a per-image bias.
 *   samplecoords [B,HW,2] float32 (normalised pixel coordinates; the 20 sin + 20 cos channels are made in the kernel,
 *                channel order of mlp2d.py:64-68)
 *   bias1  [B,256] float32 = b1 + W1[:, 0:80] . cat(camera code, identity code)
 *   w1pos  [256,48] bf16   = W1[:, 80:120] zero-padded to 48 input channels
 *   wh     [4,256,256] bf16 = weights of the four 256 -> 256 layers, nn.Linear / Conv2d layout [out][in]
 *   bh     [4,256] float32, w6 [3,256] float32, b6 [3] float32
 *   acts   [5, B*HW, 256] bf16, written when not NULL: the post-activation outputs of the five hidden layers
 *   x0     [B*HW, 48] bf16, written when not NULL: the positional encoding as the first GEMM consumed it
 *          (acts and x0: both or neither -- with neither, the inference instantiation runs, which has no store code)
 *   out    [B,3,HW] float32 (NCHW planes)
 * bf16 operands, fp32 accumulation.  Backward: dz [5, B*HW, 256] bf16 = gradients w.r.t. the five pre-activations
 * (the chain of input gradients), from grad_out [B,3,HW] and `acts`; whT holds the transposed hidden weights
 * [4,256,256] = [in][out]; colsum [5, B*ceil(HW/256), 256] float32 = column sums of dz per 256-pixel tile (tiles of one
 * image are consecutive: summed over an image they are the gradient of bias1, over everything of bh).  Weight gradients
 * are [256 x P] . [P x 256] GEMMs over (x0 | acts, dz), left to the caller's BLAS. */
int mvp_bgmlp_forward(int B, int HW, const float *samplecoords, const float *bias1, const void *w1pos, const void *wh,
                      const float *bh, const float *w6, const float *b6, void *acts /*or NULL*/, void *x0 /*or NULL*/,
                      float *out, void *stream);
int mvp_bgmlp_backward(int B, int HW, const float *grad_out, const void *acts, const void *whT, const float *w6,
                       void *dz, float *colsum, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MVP_ABI_H_ */
