/*
 * oracle/mvp_oracle.c  --  TEST INFRASTRUCTURE ONLY.  Not the product, never shipped,
 * never on the product path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.
 *
 * CPU restatement (plain C, one ray at a time) of the reference's MVP raymarch path:
 *   - ray generation              : /root/reference/extensions/utils/utils_kernel.cu:12-52
 *   - heap-BVH AABB build         : extensions/mvpraymarch/bvh.cu:157-201,
 *                                   extensions/mvpraymarch/primtransf.h:12-63
 *   - per-ray BVH traversal       : extensions/mvpraymarch/utils.h:679-685,719-815
 *   - forward march               : extensions/mvpraymarch/mvpraymarch_subset_kernel.h:7-100
 *   - backward march              : extensions/mvpraymarch/mvpraymarch_subset_kernel.h:102-216
 *   - SRT primitive transform     : extensions/mvpraymarch/primtransf.h:105-179
 *   - fade + trilinear sampler    : extensions/mvpraymarch/primsampler.h:44-91,
 *                                   extensions/mvpraymarch/utils.h:408-502,504-643
 *   - additive accumulation       : extensions/mvpraymarch/primaccum.h:37-98
 *
 * The reference's CUDA kernels cannot be built here (no nvcc, no NVIDIA device), so this
 * restatement is pinned against the reference's OWN dense PyTorch statement of the same
 * algorithm (extensions/mvpraymarch/mvpraymarch.py:553-633, the `gradcheck` oracle), executed
 * in this container in float64 by tests/golden/gen_golden.py; see tests/test_oracle_golden.py.
 *
 * Compiled twice by oracle/Makefile: -DREAL=double -> liboracle_f64.so (the checker) and
 * -DREAL=float -> liboracle_f32.so (fp32 arithmetic like the kernels; also the timed CPU
 * baseline "port" in bench.py).  A "packet" here is a single ray: the reference's per-warp
 * union hit list only adds primitives a ray never lies inside, so results are identical as
 * long as the 512-entry list does not overflow.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL double
#endif
typedef REAL real;

#define MVPO_MAXSTACK 64

static inline real rmin(real a, real b) { return (real)fmin((double)a, (double)b); }
static inline real rmax(real a, real b) { return (real)fmax((double)a, (double)b); }

#if defined(MVPO_FLOAT)
#define R_EXP(x) expf(x)
#define R_POW(x, y) powf(x, y)
#define R_FLOOR(x) floorf(x)
#define R_SQRT(x) sqrtf(x)
#define R_ABS(x) fabsf(x)
#else
#define R_EXP(x) exp(x)
#define R_POW(x, y) pow(x, y)
#define R_FLOOR(x) floor(x)
#define R_SQRT(x) sqrt(x)
#define R_ABS(x) fabs(x)
#endif

int mvpo_sizeof_real(void) { return (int)sizeof(real); }

/* Threads the OpenMP loops below run with (bench.py states it next to the CPU timing); 1 when built without OpenMP. */
#ifdef _OPENMP
#include <omp.h>
int mvpo_max_threads(void) { return omp_get_max_threads(); }
#else
int mvpo_max_threads(void) { return 1; }
#endif

/* ------------------------------------------------------------------------------------------
 * Ray generation.  utils_kernel.cu:12-52 (forward only; the reference backward kernel writes
 * nothing, utils_kernel.cu:54-95, and extensions/utils/utils.py:44-46 returns None grads).
 * pixelcoords may be NULL -> (w, h) integer grid (utils_kernel.cu:36).
 * ------------------------------------------------------------------------------------------ */
int mvpo_raydirs(int N, int H, int W, const real *campos, const real *camrot, const real *focal,
                 const real *princpt, const real *pixelcoords, real volradius, real *raypos,
                 real *raydir, real *tminmax) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n) {
        for (int h = 0; h < H; ++h) {
            for (int w = 0; w < W; ++w) {
                size_t r = ((size_t)n * H + h) * W + w;
                real o[3] = {campos[n * 3 + 0] / volradius, campos[n * 3 + 1] / volradius,
                             campos[n * 3 + 2] / volradius};
                const real *R = camrot + n * 9;
                real px = pixelcoords ? pixelcoords[r * 2 + 0] : (real)w;
                real py = pixelcoords ? pixelcoords[r * 2 + 1] : (real)h;
                real qx = (px - princpt[n * 2 + 0]) / focal[n * 2 + 0];
                real qy = (py - princpt[n * 2 + 1]) / focal[n * 2 + 1];
                real qz = (real)1;
                real d[3];
                for (int j = 0; j < 3; ++j) d[j] = R[0 + j] * qx + R[3 + j] * qy + R[6 + j] * qz;
                real inv = (real)1 / R_SQRT(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                for (int j = 0; j < 3; ++j) d[j] *= inv;
                real t1[3], t2[3];
                for (int j = 0; j < 3; ++j) {
                    t1[j] = ((real)-1 - o[j]) / d[j];
                    t2[j] = ((real)1 - o[j]) / d[j];
                }
                real tmin = rmax(rmin(t1[0], t2[0]), rmax(rmin(t1[1], t2[1]), rmin(t1[2], t2[2])));
                real tmax = rmin(rmax(t1[0], t2[0]), rmin(rmax(t1[1], t2[1]), rmax(t1[2], t2[2])));
                for (int j = 0; j < 3; ++j) {
                    raypos[r * 3 + j] = o[j];
                    raydir[r * 3 + j] = d[j];
                }
                tminmax[r * 2 + 0] = rmax(tmin, (real)0);
                tminmax[r * 2 + 1] = tmax;
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * AABB of one oriented box.  primtransf.h:12-63: corner c in {-1,+1}^3, p = c / scale,
 * world = (dot(p,R0), dot(p,R1), dot(p,R2)) + pos, with R_i the ROWS of primrot[k].
 * ------------------------------------------------------------------------------------------ */
static void leaf_aabb(const real *pos, const real *rot, const real *scale, real *mn, real *mx) {
    for (int c = 0; c < 8; ++c) {
        real p[3] = {((c & 1) ? (real)1 : (real)-1) / scale[0], ((c & 2) ? (real)1 : (real)-1) / scale[1],
                     ((c & 4) ? (real)1 : (real)-1) / scale[2]};
        for (int i = 0; i < 3; ++i) {
            real x = p[0] * rot[i * 3 + 0] + p[1] * rot[i * 3 + 1] + p[2] * rot[i * 3 + 2] + pos[i];
            if (c == 0) {
                mn[i] = x;
                mx[i] = x;
            } else {
                mn[i] = rmin(mn[i], x);
                mx[i] = rmax(mx[i], x);
            }
        }
    }
}

/* Implicit heap of 2K-1 nodes (mvpraymarch.py:57-75, fixedorder branch): internal i has
 * children 2i+1, 2i+2; leaf node K-1+k holds primitive k (identity order, mvpraymarch.py:45).
 * nodeaabb layout [N, 2K-1, 2, 3] (mvpraymarch.py:81).  bvh.cu:157-201 fills it bottom-up. */
int mvpo_aabb(int N, int K, const real *pos, const real *rot, const real *scale, real *nodeaabb) {
    int nn = 2 * K - 1;
    for (int n = 0; n < N; ++n) {
        real *A = nodeaabb + (size_t)n * nn * 6;
        for (int k = 0; k < K; ++k)
            leaf_aabb(pos + ((size_t)n * K + k) * 3, rot + ((size_t)n * K + k) * 9, scale + ((size_t)n * K + k) * 3,
                      A + (size_t)(K - 1 + k) * 6, A + (size_t)(K - 1 + k) * 6 + 3);
        for (int i = K - 2; i >= 0; --i) {
            const real *l = A + (size_t)(2 * i + 1) * 6, *r = A + (size_t)(2 * i + 2) * 6;
            for (int j = 0; j < 3; ++j) {
                A[(size_t)i * 6 + j] = rmin(l[j], r[j]);
                A[(size_t)i * 6 + 3 + j] = rmax(l[j + 3], r[j + 3]);
            }
        }
    }
    return 0;
}

/* utils.h:679-685 */
static int ray_aabb_hit_ird(const real *p0, const real *p1, const real *o, const real *ird) {
    /* max_component / min_component are fmaxf(fmaxf(x,y),z) / fminf(fminf(x,y),z) (utils.h:659-665): a NaN axis is
     * ignored, but three NaN axes give NaN and the comparison fails -- a box with NaN corners is never entered.
     * (Folding from -inf / +inf instead would answer "hit" there.) */
    real lo[3], hi[3];
    for (int j = 0; j < 3; ++j) {
        real t0 = (p0[j] - o[j]) * ird[j], t1 = (p1[j] - o[j]) * ird[j];
        lo[j] = rmin(t0, t1);
        hi[j] = rmax(t0, t1);
    }
    return rmax(rmax(lo[0], lo[1]), lo[2]) <= rmin(rmin(hi[0], hi[1]), hi[2]);
}

typedef struct {
    real xmt[3], rxmt[3], y[3];
} srt_t;

/* primtransf.h:119-132 : y = (R0*xmt.x + R1*xmt.y + R2*xmt.z) * scale */
static void srt_forward(const real *pos, const real *rot, const real *scale, const real *x, srt_t *s) {
    for (int j = 0; j < 3; ++j) s->xmt[j] = x[j] - pos[j];
    for (int j = 0; j < 3; ++j) {
        s->rxmt[j] = rot[0 + j] * s->xmt[0] + rot[3 + j] * s->xmt[1] + rot[6 + j] * s->xmt[2];
        s->y[j] = s->rxmt[j] * scale[j];
    }
}

/* Traversal for ONE ray.  utils.h:719-815 with sortboxes=false, sync (packet = this ray).
 * Returns the number of listed primitives; updates rtminmax (utils.h:757-761). */
static int traverse(int K, const real *o, const real *d, const real *A, const real *pos, const real *rot,
                    const real *scale, int maxhit, int *hits, real *rtmin, real *rtmax, long long *overflow) {
    real ird[3] = {(real)1 / d[0], (real)1 / d[1], (real)1 / d[2]};
    int stack[MVPO_MAXSTACK];
    int sp = 0, num = 0;
    stack[sp++] = -1;
    int node = 0;
    do {
        if (node >= K - 1) {
            int k = node - (K - 1);
            const real *p = pos + (size_t)k * 3, *R = rot + (size_t)k * 9, *s = scale + (size_t)k * 3;
            real xmt[3] = {o[0] - p[0], o[1] - p[1], o[2] - p[2]};
            real lo[3], hi[3];
            for (int j = 0; j < 3; ++j) { /* primtransf.h:134-153 + utils.h:747-753 */
                real r0 = (R[0 + j] * xmt[0] + R[3 + j] * xmt[1] + R[6 + j] * xmt[2]) * s[j];
                real rd = (R[0 + j] * d[0] + R[3 + j] * d[1] + R[6 + j] * d[2]) * s[j];
                real irdj = (real)1 / rd;
                real t0 = ((real)-1 - r0) * irdj, t1 = ((real)1 - r0) * irdj;
                lo[j] = rmin(t0, t1);
                hi[j] = rmax(t0, t1);
            }
            /* utils.h:752-755 with max_component / min_component (utils.h:659-665): all-NaN -> NaN -> no hit */
            const real tn = rmax(rmax(lo[0], lo[1]), lo[2]), tf = rmin(rmin(hi[0], hi[1]), hi[2]);
            if (tn <= tf) {
                *rtmin = rmin(*rtmin, tn);
                *rtmax = rmax(*rtmax, tf);
                if (num < maxhit)
                    hits[num++] = k;
                else if (overflow)
                    ++*overflow;
            }
            node = stack[--sp];
        } else {
            int cl = 2 * node + 1, cr = 2 * node + 2;
            int tl = ray_aabb_hit_ird(A + (size_t)cl * 6, A + (size_t)cl * 6 + 3, o, ird);
            int tr = ray_aabb_hit_ird(A + (size_t)cr * 6, A + (size_t)cr * 6 + 3, o, ird);
            if (!tl && !tr) {
                node = stack[--sp];
            } else {
                node = tl ? cl : cr;
                if (tl && tr) stack[sp++] = cr;
            }
        }
    } while (node != -1);
    return num;
}

typedef struct {
    int i0[3];    /* floor(ix), floor(iy), floor(iz)  (x->W, y->H, z->D) */
    real f[3];    /* ix, iy, iz                                             */
    real w[8];    /* trilinear weights, corner c: bit0 -> +x, bit1 -> +y, bit2 -> +z */
    int idx[8];   /* voxel index ((z*TH)+y)*TW+x or -1 when out of bounds (zero padding) */
} tri_t;

/* utils.h:408-468 (align_corners=True, clamp to +-100, zero padding through bounds checks) */
static void tri_setup(int TD, int TH, int TW, const real *y, tri_t *t) {
    int T[3] = {TW, TH, TD};
    for (int j = 0; j < 3; ++j) {
        real u = rmax((real)-100, rmin((real)100, (y[j] + (real)1) / (real)2)) * (real)(T[j] - 1);
        t->f[j] = u;
        t->i0[j] = (int)R_FLOOR(u);
    }
    for (int c = 0; c < 8; ++c) {
        int cx = c & 1, cy = (c >> 1) & 1, cz = (c >> 2) & 1;
        int x = t->i0[0] + cx, yy = t->i0[1] + cy, z = t->i0[2] + cz;
        real wx = cx ? (t->f[0] - (real)t->i0[0]) : ((real)(t->i0[0] + 1) - t->f[0]);
        real wy = cy ? (t->f[1] - (real)t->i0[1]) : ((real)(t->i0[1] + 1) - t->f[1]);
        real wz = cz ? (t->f[2] - (real)t->i0[2]) : ((real)(t->i0[2] + 1) - t->f[2]);
        t->w[c] = wx * wy * wz;
        t->idx[c] = (x >= 0 && x < TW && yy >= 0 && yy < TH && z >= 0 && z < TD) ? ((z * TH) + yy) * TW + x : -1;
    }
}

static inline int srt_valid(const real *y) { /* primtransf.h:112-117, strict */
    return y[0] > (real)-1 && y[0] < (real)1 && y[1] > (real)-1 && y[1] < (real)1 && y[2] > (real)-1 &&
           y[2] < (real)1;
}

static inline real fade_of(const real *y, real fadescale, real fadeexp) { /* primsampler.h:48-51 */
    return R_EXP(-fadescale *
                 (R_POW(R_ABS(y[0]), fadeexp) + R_POW(R_ABS(y[1]), fadeexp) + R_POW(R_ABS(y[2]), fadeexp)));
}

/* ------------------------------------------------------------------------------------------
 * Forward march.  mvpraymarch_subset_kernel.h:7-100.  raysat may be NULL (no-grad mode,
 * mvpraymarch.py:147-152).  stats (may be NULL): [0] rays with >=1 hit, [1] sum of list lengths,
 * [2] evaluated samples, [3] list overflows, [4] march steps, [5] saturated rays.
 * ------------------------------------------------------------------------------------------ */
/* trilinear lookup of a C-channel channels-last grid at box coordinate y (utils.h:408-502); zero padding */
static void tri_fetch(const real *G, int C, const tri_t *tr, real *out) {
    for (int ch = 0; ch < C; ++ch) out[ch] = 0;
    for (int c = 0; c < 8; ++c)
        if (tr->idx[c] >= 0)
            for (int ch = 0; ch < C; ++ch) out[ch] += G[(size_t)tr->idx[c] * C + ch] * tr->w[c];
}

/* Optional per-ray diagnostics of the forward march (tests only; all NULL by default, set through
 * mvpo_set_ray_diagnostics and valid for the next calls until reset):
 *   margin[r]   = min over the evaluated samples of |alpha_after_sample - 1|  (INFINITY when the ray takes no sample):
 *                 how close the saturation decision `newalpha >= 1` (primaccum.h:71) came to flipping;
 *   hitcount[r] = primitives listed for the ray (utils.h:757-781);   nsamples[r] = samples evaluated;
 *   edge[r]     = the largest opacity increment |alpha * stepsize| among the (step, primitive) pairs whose INCLUSION came
 *                 within eps of flipping: the strict box test (primtransf.h:112-117) with max|y_i| within eps * max|scale_i|
 *                 of 1 (eps in world units: what fp32 position round-off can move a sample), or the march bound
 *                 `t < tmax + 1e-5` (subset_kernel.h:76) within eps.  0 when no decision was close.  Like saturation these
 *                 are discontinuities of the rendering in the sample position: an fp32 march may decide them differently
 *                 from this float64 one, and then differs by that increment. */
static real *g_margin = NULL, *g_edge = NULL;
static real g_edge_eps = 0;
static int *g_hitcount = NULL, *g_nsamples = NULL;
void mvpo_set_ray_diagnostics(real *margin, int *hitcount, int *nsamples) {
    g_margin = margin;
    g_hitcount = hitcount;
    g_nsamples = nsamples;
}
void mvpo_set_edge_diagnostics(real *edge, real eps) {
    g_edge = edge;
    g_edge_eps = eps;
}

/* |opacity increment| of a sample taken at box coordinate y pulled just inside the box (edge diagnostic only) */
static real edge_increment(const real *y, const real *Tk, int TD, int TH, int TW, const real *Wk, int WD, int WH, int WW,
                           real fadescale, real fadeexp, real stepsize) {
    real yc[3];
    for (int j = 0; j < 3; ++j) yc[j] = rmin(rmax(y[j], (real)-0.999999), (real)0.999999);
    real y1c[3] = {yc[0], yc[1], yc[2]};
    if (Wk) {
        tri_t tw;
        tri_setup(WD, WH, WW, yc, &tw);
        tri_fetch(Wk, 3, &tw, y1c);
    }
    tri_t trc;
    tri_setup(TD, TH, TW, y1c, &trc);
    real vc[4];
    tri_fetch(Tk, 4, &trc, vc);
    return R_ABS(vc[3] * fade_of(yc, fadescale, fadeexp) * stepsize);
}

/* warp may be NULL (algo 0).  With a warp field [N,K,WD,WH,WW,3] the template is sampled at y1 = warp(y0)
 * (PrimSamplerTW<true>, primsampler.h:53-58); the fade still uses y0. */
int mvpo_march_forward(int N, int H, int W, int K, const real *raypos, const real *raydir, real stepsize,
                       const real *tminmax, const real *nodeaabb, const real *primpos, const real *primrot,
                       const real *primscale, int TD, int TH, int TW, const real *tplate, int WD, int WH, int WW,
                       const real *warp, real *rayrgba, real *raysat, real fadescale, real fadeexp, int maxhitboxes,
                       long long *stats) {
    const int nn = 2 * K - 1;
    const size_t V = (size_t)TD * TH * TW;
    const size_t VW = (size_t)WD * WH * WW;
    long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0;
    if (maxhitboxes <= 0) maxhitboxes = 512;
#pragma omp parallel reduction(+ : st0, st1, st2, st3, st4, st5)
    {
        int *hits = (int *)malloc(sizeof(int) * (size_t)maxhitboxes);
#pragma omp for schedule(dynamic, 64)
        for (long long r = 0; r < (long long)N * H * W; ++r) {
            int n = (int)(r / ((long long)H * W));
            const real *o = raypos + r * 3, *d = raydir + r * 3;
            const real tmn = tminmax[r * 2 + 0], tmx = tminmax[r * 2 + 1];
            const real *A = nodeaabb + (size_t)n * nn * 6;
            const real *pp = primpos + (size_t)n * K * 3, *pr = primrot + (size_t)n * K * 9,
                       *ps = primscale + (size_t)n * K * 3;
            const real *T = tplate + (size_t)n * K * V * 4;
            real rtmin = INFINITY, rtmax = -INFINITY;
            long long ovf = 0;
            int nh = traverse(K, o, d, A, pp, pr, ps, maxhitboxes, hits, &rtmin, &rtmax, &ovf);
            st3 += ovf;
            real rgba[4] = {0, 0, 0, 0}, sat3[3] = {-1, -1, -1};
            int sat = 0, myns = 0;
            real margin = INFINITY, edge = 0;
            rtmin = rmax(rtmin, tmn); /* subset_kernel.h:63-64 */
            rtmax = rmin(rtmax, tmx);
            if (nh > 0 && rtmin < INFINITY) {
                st0 += 1;
                st1 += nh;
                real t = tmn; /* subset_kernel.h:67-72 */
                real x[3] = {o[0] + d[0] * tmn, o[1] + d[1] * tmn, o[2] + d[2] * tmn};
                real incs = R_FLOOR((rtmin - t) / stepsize);
                t += incs * stepsize;
                for (int j = 0; j < 3; ++j) x[j] += d[j] * incs * stepsize;
                while (!(t > rtmax + (real)1e-5 || sat)) { /* subset_kernel.h:76 */
                    st4 += 1;
                    for (int ks = 0; ks < nh; ++ks) {
                        int k = hits[ks];
                        srt_t s;
                        srt_forward(pp + (size_t)k * 3, pr + (size_t)k * 9, ps + (size_t)k * 3, x, &s);
                        if (g_edge && !sat) { /* diagnostics only: was the inclusion of this pair a close call? */
                            const real *sc = ps + (size_t)k * 3;
                            real cheb = rmax(R_ABS(s.y[0]), rmax(R_ABS(s.y[1]), R_ABS(s.y[2])));
                            real smax = rmax(R_ABS(sc[0]), rmax(R_ABS(sc[1]), R_ABS(sc[2])));
                            int near_box = R_ABS(cheb - (real)1) < g_edge_eps * smax;
                            int near_t = R_ABS(t - (rtmax + (real)1e-5)) < g_edge_eps && cheb < (real)1 + g_edge_eps * smax;
                            if (near_box || near_t)
                                edge = rmax(edge, edge_increment(s.y, T + (size_t)k * V * 4, TD, TH, TW,
                                                                 warp ? warp + ((size_t)n * K + k) * VW * 3 : NULL, WD, WH,
                                                                 WW, fadescale, fadeexp, stepsize));
                        }
                        if (srt_valid(s.y) && !sat && t < rtmax + (real)1e-5) {
                            st2 += 1;
                            real fade = fade_of(s.y, fadescale, fadeexp);
                            real y1[3] = {s.y[0], s.y[1], s.y[2]};
                            if (warp) {
                                tri_t tw;
                                tri_setup(WD, WH, WW, s.y, &tw);
                                tri_fetch(warp + ((size_t)n * K + k) * VW * 3, 3, &tw, y1);
                            }
                            tri_t tr;
                            tri_setup(TD, TH, TW, y1, &tr);
                            real v[4];
                            const real *Tk = T + (size_t)k * V * 4;
                            tri_fetch(Tk, 4, &tr, v);
                            v[3] *= fade;
                            /* primaccum.h:63-79 */
                            real newalpha = rgba[3] + v[3] * stepsize;
                            real contrib = rmin(newalpha, (real)1) - rgba[3];
                            margin = rmin(margin, R_ABS(newalpha - (real)1));
                            ++myns;
                            rgba[0] += v[0] * contrib;
                            rgba[1] += v[1] * contrib;
                            rgba[2] += v[2] * contrib;
                            rgba[3] += contrib;
                            if (newalpha >= (real)1) {
                                if (!sat) {
                                    sat3[0] = v[0];
                                    sat3[1] = v[1];
                                    sat3[2] = v[2];
                                }
                                sat = 1;
                            }
                        }
                    }
                    t += stepsize; /* subset_kernel.h:95-96: incremental adds */
                    for (int j = 0; j < 3; ++j) x[j] += d[j] * stepsize;
                }
                /* diagnostics only: the first step the march bound EXCLUDED, if it was excluded by less than eps (a box
                 * whose exit sits that close behind the step would have included it) */
                if (g_edge && !sat && t - (rtmax + (real)1e-5) < g_edge_eps) {
                    for (int ks = 0; ks < nh; ++ks) {
                        int k = hits[ks];
                        const real *sc = ps + (size_t)k * 3;
                        srt_t s;
                        srt_forward(pp + (size_t)k * 3, pr + (size_t)k * 9, sc, x, &s);
                        real cheb = rmax(R_ABS(s.y[0]), rmax(R_ABS(s.y[1]), R_ABS(s.y[2])));
                        real smax = rmax(R_ABS(sc[0]), rmax(R_ABS(sc[1]), R_ABS(sc[2])));
                        if (cheb < (real)1 + g_edge_eps * smax)
                            edge = rmax(edge, edge_increment(s.y, T + (size_t)k * V * 4, TD, TH, TW,
                                                             warp ? warp + ((size_t)n * K + k) * VW * 3 : NULL, WD, WH,
                                                             WW, fadescale, fadeexp, stepsize));
                    }
                }
            }
            st5 += sat;
            if (g_margin) g_margin[r] = margin;
            if (g_edge) g_edge[r] = edge;
            if (g_hitcount) g_hitcount[r] = nh;
            if (g_nsamples) g_nsamples[r] = myns;
            for (int j = 0; j < 4; ++j) rayrgba[r * 4 + j] = rgba[j];
            if (raysat)
                for (int j = 0; j < 3; ++j) raysat[r * 3 + j] = sat3[j];
        }
        free(hits);
    }
    if (stats) {
        stats[0] = st0;
        stats[1] = st1;
        stats[2] = st2;
        stats[3] = st3;
        stats[4] = st4;
        stats[5] = st5;
    }
    return 0;
}

static inline void atomic_add(real *p, real v) {
#pragma omp atomic
    *p += v;
}

/* ------------------------------------------------------------------------------------------
 * Backward march.  mvpraymarch_subset_kernel.h:102-216 (forwarddir=true): same traversal, same
 * forward-order re-march with the running RGBA recomputed; primaccum.h:81-98,
 * primsampler.h:68-91, utils.h:504-643, primtransf.h:155-179.  The grad buffers are
 * ACCUMULATED INTO (caller zero-fills, as mvpraymarch.py:240-246 does).
 * ------------------------------------------------------------------------------------------ */
int mvpo_march_backward(int N, int H, int W, int K, const real *raypos, const real *raydir, real stepsize,
                        const real *tminmax, const real *nodeaabb, const real *primpos, const real *primrot,
                        const real *primscale, int TD, int TH, int TW, const real *tplate, int WD, int WH, int WW,
                        const real *warp, const real *raysat, const real *grad_rayrgba, real *grad_primpos,
                        real *grad_primrot, real *grad_primscale, real *grad_tplate, real *grad_warp, real fadescale,
                        real fadeexp, int maxhitboxes) {
    const int nn = 2 * K - 1;
    const size_t V = (size_t)TD * TH * TW;
    const size_t VW = (size_t)WD * WH * WW;
    if (maxhitboxes <= 0) maxhitboxes = 512;
#pragma omp parallel
    {
        int *hits = (int *)malloc(sizeof(int) * (size_t)maxhitboxes);
#pragma omp for schedule(dynamic, 64)
        for (long long r = 0; r < (long long)N * H * W; ++r) {
            int n = (int)(r / ((long long)H * W));
            const real *o = raypos + r * 3, *d = raydir + r * 3;
            const real tmn = tminmax[r * 2 + 0], tmx = tminmax[r * 2 + 1];
            const real *A = nodeaabb + (size_t)n * nn * 6;
            const real *pp = primpos + (size_t)n * K * 3, *pr = primrot + (size_t)n * K * 9,
                       *ps = primscale + (size_t)n * K * 3;
            real *gpp = grad_primpos + (size_t)n * K * 3, *gpr = grad_primrot + (size_t)n * K * 9,
                 *gps = grad_primscale + (size_t)n * K * 3;
            const real *T = tplate + (size_t)n * K * V * 4;
            real *gT = grad_tplate + (size_t)n * K * V * 4;
            const real *dL = grad_rayrgba + r * 4; /* primaccum.h:58-61 */
            const real *rs = raysat + r * 3;
            real rtmin = INFINITY, rtmax = -INFINITY;
            int nh = traverse(K, o, d, A, pp, pr, ps, maxhitboxes, hits, &rtmin, &rtmax, NULL);
            rtmin = rmax(rtmin, tmn);
            rtmax = rmin(rtmax, tmx);
            if (!(nh > 0 && rtmin < INFINITY)) continue;
            real rgba[4] = {0, 0, 0, 0};
            int sat = 0;
            real t = tmn;
            real x[3] = {o[0] + d[0] * tmn, o[1] + d[1] * tmn, o[2] + d[2] * tmn};
            real incs = R_FLOOR((rtmin - t) / stepsize);
            t += incs * stepsize;
            for (int j = 0; j < 3; ++j) x[j] += d[j] * incs * stepsize;
            while (t < rtmax + (real)1e-5 && !sat) { /* subset_kernel.h:181-184 */
                for (int ks = 0; ks < nh; ++ks) {
                    int k = hits[ks];
                    const real *R = pr + (size_t)k * 9, *sc = ps + (size_t)k * 3;
                    srt_t s;
                    srt_forward(pp + (size_t)k * 3, R, sc, x, &s);
                    if (!(srt_valid(s.y) && !sat && t < rtmax + (real)1e-5)) continue;
                    real fade = fade_of(s.y, fadescale, fadeexp);
                    real y1[3] = {s.y[0], s.y[1], s.y[2]};
                    tri_t tw;
                    const real *Wk = NULL;
                    if (warp) {
                        Wk = warp + ((size_t)n * K + k) * VW * 3;
                        tri_setup(WD, WH, WW, s.y, &tw);
                        tri_fetch(Wk, 3, &tw, y1);
                    }
                    tri_t tr;
                    tri_setup(TD, TH, TW, y1, &tr);
                    const real *Tk = T + (size_t)k * V * 4;
                    real *gTk = gT + (size_t)k * V * 4;
                    real v[4];
                    tri_fetch(Tk, 4, &tr, v);
                    v[3] *= fade; /* sample.w (primsampler.h:63) */
                    /* primaccum.h:81-98 */
                    real a = v[3] * stepsize;
                    int thissat = rgba[3] + a >= (real)1;
                    sat = sat || thissat;
                    real weight = sat ? ((real)1 - rgba[3]) : a;
                    real dLs[4];
                    dLs[0] = weight * dL[0];
                    dLs[1] = weight * dL[1];
                    dLs[2] = weight * dL[2];
                    if (sat) {
                        dLs[3] = 0;
                    } else {
                        int has = rs[0] > (real)-1;
                        real q0 = v[0] - (has ? rs[0] : 0), q1 = v[1] - (has ? rs[1] : 0),
                             q2 = v[2] - (has ? rs[2] : 0), q3 = (real)1 - (has ? (real)1 : 0);
                        dLs[3] = stepsize * (q0 * dL[0] + q1 * dL[1] + q2 * dL[2] + q3 * dL[3]);
                    }
                    rgba[0] += v[0] * weight;
                    rgba[1] += v[1] * weight;
                    rgba[2] += v[2] * weight;
                    rgba[3] += weight;
                    /* primsampler.h:70-76 */
                    real gy[3];
                    for (int j = 0; j < 3; ++j) {
                        real sg = s.y[j] > 0 ? (real)1 : (real)-1;
                        real dfade = -(fadescale * fadeexp) * R_POW(R_ABS(s.y[j]), fadeexp - (real)1) * sg;
                        gy[j] = dfade * v[3] * dLs[3];
                    }
                    dLs[3] *= fade;
                    /* utils.h:582-642 on the template at y1: scatter to the 8 corners, then d/d(y1) */
                    real g1[3];
                    {
                        real gi[3] = {0, 0, 0};
                        for (int c = 0; c < 8; ++c) {
                            if (tr.idx[c] < 0) continue;
                            real dot = 0;
                            for (int ch = 0; ch < 4; ++ch) {
                                atomic_add(gTk + (size_t)tr.idx[c] * 4 + ch, tr.w[c] * dLs[ch]);
                                dot += Tk[(size_t)tr.idx[c] * 4 + ch] * dLs[ch];
                            }
                            int cx = c & 1, cy = (c >> 1) & 1, cz = (c >> 2) & 1;
                            real wx = cx ? (tr.f[0] - (real)tr.i0[0]) : ((real)(tr.i0[0] + 1) - tr.f[0]);
                            real wy = cy ? (tr.f[1] - (real)tr.i0[1]) : ((real)(tr.i0[1] + 1) - tr.f[1]);
                            real wz = cz ? (tr.f[2] - (real)tr.i0[2]) : ((real)(tr.i0[2] + 1) - tr.f[2]);
                            gi[0] += (cx ? (real)1 : (real)-1) * wy * wz * dot;
                            gi[1] += (cy ? (real)1 : (real)-1) * wx * wz * dot;
                            gi[2] += (cz ? (real)1 : (real)-1) * wx * wy * dot;
                        }
                        g1[0] = (real)(TW - 1) / (real)2 * gi[0];
                        g1[1] = (real)(TH - 1) / (real)2 * gi[1];
                        g1[2] = (real)(TD - 1) / (real)2 * gi[2];
                    }
                    if (warp) { /* primsampler.h:82-85: scatter dL_y1 into the warp field, chain to y0 */
                        real *gWk = grad_warp + ((size_t)n * K + k) * VW * 3;
                        real gi[3] = {0, 0, 0};
                        for (int c = 0; c < 8; ++c) {
                            if (tw.idx[c] < 0) continue;
                            real dot = 0;
                            for (int ch = 0; ch < 3; ++ch) {
                                atomic_add(gWk + (size_t)tw.idx[c] * 3 + ch, tw.w[c] * g1[ch]);
                                dot += Wk[(size_t)tw.idx[c] * 3 + ch] * g1[ch];
                            }
                            int cx = c & 1, cy = (c >> 1) & 1, cz = (c >> 2) & 1;
                            real wx = cx ? (tw.f[0] - (real)tw.i0[0]) : ((real)(tw.i0[0] + 1) - tw.f[0]);
                            real wy = cy ? (tw.f[1] - (real)tw.i0[1]) : ((real)(tw.i0[1] + 1) - tw.f[1]);
                            real wz = cz ? (tw.f[2] - (real)tw.i0[2]) : ((real)(tw.i0[2] + 1) - tw.f[2]);
                            gi[0] += (cx ? (real)1 : (real)-1) * wy * wz * dot;
                            gi[1] += (cy ? (real)1 : (real)-1) * wx * wz * dot;
                            gi[2] += (cz ? (real)1 : (real)-1) * wx * wy * dot;
                        }
                        gy[0] += (real)(WW - 1) / (real)2 * gi[0];
                        gy[1] += (real)(WH - 1) / (real)2 * gi[1];
                        gy[2] += (real)(WD - 1) / (real)2 * gi[2];
                    } else {
                        gy[0] += g1[0];
                        gy[1] += g1[1];
                        gy[2] += g1[2];
                    }
                    /* primtransf.h:155-179 */
                    real g2[3];
                    for (int j = 0; j < 3; ++j) {
                        atomic_add(gps + (size_t)k * 3 + j, s.rxmt[j] * gy[j]);
                        g2[j] = gy[j] * sc[j];
                    }
                    for (int i = 0; i < 3; ++i) {
                        for (int j = 0; j < 3; ++j) atomic_add(gpr + (size_t)k * 9 + i * 3 + j, s.xmt[i] * g2[j]);
                        atomic_add(gpp + (size_t)k * 3 + i, -(R[i * 3 + 0] * g2[0] + R[i * 3 + 1] * g2[1] + R[i * 3 + 2] * g2[2]));
                    }
                }
                t += stepsize;
                for (int j = 0; j < 3; ++j) x[j] += d[j] * stepsize;
            }
        }
        free(hits);
    }
    return 0;
}
