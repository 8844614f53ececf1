"""Shared helpers of the GPU parity tests (test infrastructure)."""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN


def to_dev(x, dev="cuda"):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).to(dev).contiguous()


def npf(t):
    return t.detach().cpu().numpy().astype(np.float64)


def cosine(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    den = np.sqrt((a * a).sum() * (b * b).sum())
    return 1.0 if den == 0 else float((a * b).sum() / den)


def load_krt_400940():
    """Camera 400940 exactly as the reference's fixture derives it (tests/test_extensions.py:25-41,
    utils.py:142-170): intrin = K^T, extrin = T[:4,:3]^T, campos = -R^T t."""
    item = json.load(open(os.path.join(GOLDEN, "camera_400940.json")))["KRT"][0]
    intrin = np.array(item["K"]).T
    extrin = np.array(item["T"])[:4, :3].T
    campos = (-np.dot(extrin[:3, :3].T, extrin[:3, 3])).astype(np.float32)[None]
    camrot = extrin[:3, :3].astype(np.float32)[None]
    focal = np.diag(intrin[:2, :2]).astype(np.float32)[None]
    princpt = intrin[:2, 2].astype(np.float32)[None]
    return campos, camrot, focal, princpt


SAT_BAND = 1e-4  # a ray may disagree with the float64 oracle on its saturating sample only inside this band
HIT_FRAC = 1e-3  # ... on at most this fraction of the rays that hit anything (FragileRays.bound)
MASK_RECORDS = []  # one entry per FragileRays use: written to gpurun_out/parity_masks.json at the end of a session (conftest.py)
SAT_ROUNDOFF = 2e-6  # ... and a ray whose alpha passes 1 by less than THIS (fp32 round-off of a sum of ~10^2 increments) is fragile whether or
#                      not its raysat shows it: at fine steps the neighbouring sample has the same colour to 1e-3 (fuzz seed 4093, see FragileRays)
EDGE_JUMP = 5e-5  # ... and on including a sample that sits on a box face / the march bound (oracle `edge`, see FragileRays)


class FragileRays:
    """Saturation (primaccum.h:71, `newalpha >= 1`) is a discontinuity of the gradient: a ray whose running alpha
    passes 1.0 by less than fp32 round-off may saturate at a different sample than in float64.  Such rays get zero
    upstream gradient on both sides -- but ONLY when the ORACLE says they are borderline: `margin` is the float64
    oracle's own min over the ray's samples of |alpha_after_sample - 1| (Oracle.march_forward(ray_diagnostics=True)).
    A ray whose kernel `raysat` disagrees with the oracle's although its margin is >= SAT_BAND fails the test.

    The other discontinuity is the INCLUSION of a sample: the strict box test and the march bound.  `edge` (optional;
    Oracle.march_forward(ray_diagnostics=True)["edge"]) is the float64 oracle's own largest opacity increment among
    the inclusion decisions that came within fp32 position round-off of flipping.  Rays with edge > EDGE_JUMP are
    masked like the saturation-fragile ones (an fp32 march -- the reference's included -- may decide them either
    way); that matters with fade parameters that leave a visible opacity AT the box faces (fadescale well below 8) or
    very opaque slabs, and is nil for the reference's fade(8, 8) at ordinary opacities (e^-8 at the face)."""

    def __init__(self, ref_sat, margin, gout, max_frac=None, min_allowed=2, edge=None, nsamples=None, label=None,
                 edge_jump=None, sat_roundoff=None):
        self.ref_sat, self.margin, self.gout = ref_sat, margin, gout
        self.max_frac, self.min_allowed = max_frac, min_allowed
        self.sat_roundoff = sat_roundoff  # (the fuzz passes SAT_ROUNDOFF: see __call__)
        # `edge_jump`: a tighter threshold than EDGE_JUMP where the scene calls for one -- see edge_jump_for()
        self.edge_mask = None if edge is None else np.asarray(edge) > (EDGE_JUMP if edge_jump is None else min(EDGE_JUMP, edge_jump))
        self.hits = None if nsamples is None else int((np.asarray(nsamples) > 0).sum())
        self.label = label or os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]   # (pytest names the running test)
        self.mask = None

    def bound(self, size):
        """Rays that may be masked for saturating elsewhere than the oracle: with the oracle's per-ray sample counts, 1e-3 of
        the rays that HIT anything (round 5; measured 0-19 rays per BASELINE-size scene); an explicit `max_frac` (scenes built
        to sit on the discontinuity: signed opacity, fade parameters with opacity at the box faces) or no sample counts: that
        fraction (default 5e-3) of all rays."""
        if self.hits is not None and self.max_frac is None:
            return max(self.min_allowed, HIT_FRAC * self.hits)
        return max(self.min_allowed, (self.max_frac or 0.005) * size)

    def __call__(self, hip_raysat):
        diff = np.abs(hip_raysat - self.ref_sat).max(-1) > 1e-3 * max(1.0, np.abs(self.ref_sat).max())
        if self.edge_mask is not None:  # a ray that included one more / one fewer sample may also saturate elsewhere
            diff = diff & ~self.edge_mask
        # The comparison above SEES a ray that saturated at another sample only when that sample's colour differs.  With fine
        # steps it does not (dt = 1/512: neighbouring samples agree to 1e-3), yet the alpha gradient of the two samples swaps
        # roles: fuzz seed 4093 (K = 1, 147 samples per ray) -- ONE ray whose alpha passes 1.0 by 3.2e-7 in float64 carries the
        # whole 1.6e-3 error of the slab gradient, identically in every backward owner (tools/diag_pose_localize.py 4093
        # template).  A margin inside fp32 round-off of the alpha sum is fragile by the oracle's own account.
        if self.sat_roundoff is not None:
            diff = diff | (np.asarray(self.margin) < self.sat_roundoff)
        unjustified = diff & ~(self.margin < SAT_BAND)
        assert unjustified.sum() == 0, ("rays saturate differently from the oracle outside the %g band" % SAT_BAND,
                                        int(unjustified.sum()), float(self.margin[unjustified].min()))
        MASK_RECORDS.append({"config": self.label, "rays": int(diff.size), "hitting_rays": self.hits,
                             "masked_saturation": int(diff.sum()),
                             "masked_edge": 0 if self.edge_mask is None else int(self.edge_mask.sum()),
                             "bound": float(self.bound(diff.size))})
        assert diff.sum() <= self.bound(diff.size), (int(diff.sum()), self.bound(diff.size), self.label)
        self.mask = diff if self.edge_mask is None else (diff | self.edge_mask)
        return self.masked()

    def masked(self):
        g = self.gout.copy()
        g[self.mask] = 0.0
        return g


def edge_jump_for(fwd_tol_abs, template):
    """The opacity increment at which ONE flipped inclusion decision moves a ray's colour by half the forward tolerance:
    a flipped sample changes rgb by (increment x the sample's colour), so the threshold on the oracle's `edge` must follow
    the ratio tolerance / largest slab colour.  EDGE_JUMP (5e-5) is that ratio for opaque scenes (max |rgba| ~ 255 x alpha);
    a nearly transparent image of bright slabs has a tolerance 2e-4 x (a small max |rgba|) against the same colours --
    fuzz seed 1847 (opacity x 0.5, fade(3.6, 2.6)): one ray, edge 4.96e-5, off by 0.0054 against 0.0052 while the oracle's
    own fp32 build misses four other rays by 0.003 the same way.  Named by the float64 oracle, like every excused ray."""
    cmax = float(np.abs(np.asarray(template)[..., :3]).max())
    return 0.5 * float(fwd_tol_abs) / max(cmax, 1e-30)


def scene_rays(oracle, s):
    """Rays of a synthetic scene computed by the oracle (float64)."""
    return oracle.raydirs(s["campos"].numpy(), s["camrot"].numpy(), s["focal"].numpy(), s["princpt"].numpy(),
                          s["pixelcoords"].numpy(), s["volradius"])


def make_placement_inputs(nprims, B=None, V=7306, T=1024, seed=77):
    """Seeded inputs of the primitive-placement tests (shared by tests/golden/gen_placement.py, which feeds them to the
    reference's own statements): geo [B,V,3] in millimetre-like units, idxim [T,T,3] vertex indices, barim [T,T,3]
    barycentric weights, volradius, and three weight arrays for a scalar loss."""
    rng = np.random.default_rng(seed + nprims)
    B = B or (2 if nprims == 16384 else 3)
    geo = (rng.normal(size=(B, V, 3)) * 60.0).astype(np.float32)
    idxim = rng.integers(0, V, size=(T, T, 3), dtype=np.int64)
    bar = rng.random(size=(T, T, 3)).astype(np.float32) + np.float32(0.05)
    barim = (bar / bar.sum(-1, keepdims=True)).astype(np.float32)
    ny, nx = (16, 16) if nprims == 256 else (128, 128)
    w = (rng.normal(size=(B, nprims, 3)).astype(np.float32), rng.normal(size=(B, ny, nx, 3)).astype(np.float32),
         rng.normal(size=(B, ny, nx, 3)).astype(np.float32))
    return geo, idxim, barim, 256.0, w


# ---- degenerate / non-finite primitive inputs (tests/test_gpu_hardening.py, tests/test_oracle_degenerate.py) -------------
DEGENERATE_CASES = ["scale0_some", "scale0_all", "scale0_axis_aligned", "scale_inf_some", "nan_alpha_slab", "nan_rgb_voxel",
                    "inf_alpha_voxel", "nan_primpos", "inf_primpos", "nan_primrot", "tmin_eq_tmax"]


def degenerate_case(name, oracle):
    """Inputs of one degenerate-input scene (numpy, float32 values), as the reference's callers can produce them:

      scale0_*        a fresh DecoderAssembler has adaptwarps = 0 (models/decoders/assembler.py:66,199) => primscale = 0 =>
                      1/scale = inf in the AABB corners (primtransf.h:12-63) unless running_avg_scale ran first
                      (ddp-train.py:374-377).  With a general rotation the leaf box becomes (-inf, +inf)^3, every ray
                      "hits" the primitive (r0 = rd = 0 -> slab interval (-inf, +inf), utils.h:744-755) and samples its
                      centre voxel cell at EVERY lattice step of the ray; with an axis-aligned rotation the corners are
                      inf * 0 = NaN, the box is NaN and the reference never enters it (utils.h:659-665,679-685);
      scale_inf_some  box-space coordinates are +-inf / NaN: never hit;
      nan_* / inf_*   what a diverged decoder hands over (ddp-train.py:436-439,469-472 exist because it happens): NaN / Inf in
                      the slab of ONE visible primitive, in one primitive's position / rotation;
      tmin_eq_tmax    rays whose march interval is a single point: one lattice step at most (subset_kernel.h:66-72,84).
    Returns a dict of arrays + stepsize; `poisoned` lists the primitives touched."""
    from ava256_amd.scene import make_scene
    N, H, W, K = (1, 24, 24, 16) if name == "scale0_all" else (1, 40, 40, 64)
    s = make_scene(N, H, W, K, device="cpu", seed=5, alpha_gain=1.0)
    rp, rd, tm = scene_rays(oracle, s)
    a = {k: s[k].numpy().copy() for k in ("primpos", "primrot", "primscale", "template")}
    vis = [9, 27, 36]   # primitives the front cameras see (asserted by the tests through the clean gradient)
    one = vis[1]
    poisoned = []
    if name == "scale0_some":
        a["primscale"][0, vis] = 0.0
        poisoned = vis
    elif name == "scale0_all":
        a["primscale"][:] = 0.0
        poisoned = list(range(K))
    elif name == "scale0_axis_aligned":
        a["primscale"][0, vis] = 0.0
        a["primrot"][0, vis] = np.eye(3, dtype=np.float32)
        poisoned = vis
    elif name == "scale_inf_some":
        a["primscale"][0, vis] = np.inf
        poisoned = vis
    elif name == "nan_alpha_slab":
        a["template"][0, one, ..., 3] = np.nan
        poisoned = [one]
    elif name == "nan_rgb_voxel":
        a["template"][0, one, 3:5, 3:5, 3:5, 1] = np.nan
        poisoned = [one]
    elif name == "inf_alpha_voxel":
        a["template"][0, one, 3:5, 3:5, 3:5, 3] = np.inf
        poisoned = [one]
    elif name == "nan_primpos":
        a["primpos"][0, one, 1] = np.nan
        poisoned = [one]
    elif name == "inf_primpos":
        a["primpos"][0, one, 1] = np.inf
        poisoned = [one]
    elif name == "nan_primrot":
        a["primrot"][0, one, 1, 2] = np.nan
        poisoned = [one]
    elif name == "tmin_eq_tmax":
        t0 = tm[..., 0] + 0.5 * (tm[..., 1] - tm[..., 0])   # a point in the middle of the volume: inside the shell for most rays
        tm = np.stack([t0, t0], -1)
    else:
        raise KeyError(name)
    return dict(raypos=rp, raydir=rd, tminmax=tm, stepsize=s["stepsize"], poisoned=poisoned, visible=vis, **a)


def assert_same_poison(got, ref, tol, what):
    """`got` (kernel) against `ref` (float64 oracle) where the reference's arithmetic yields non-finite values: the SAME
    elements are non-finite (NaN where NaN, +-inf where +-inf), every other element within tol * max |finite ref|."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    nan_r, nan_g = np.isnan(ref), np.isnan(got)
    assert (nan_r == nan_g).all(), (what, "NaN pattern differs", int(nan_r.sum()), int(nan_g.sum()), int((nan_r != nan_g).sum()))
    inf_r, inf_g = np.isinf(ref), np.isinf(got)
    assert (inf_r == inf_g).all() and (np.sign(ref[inf_r]) == np.sign(got[inf_g])).all(), (what, "Inf pattern differs")
    fin = np.isfinite(ref)
    if fin.any():
        scale = max(np.abs(ref[fin]).max(), 1e-30)
        err = np.abs(got[fin] - ref[fin]).max()
        assert err <= tol * scale, (what, err, scale)
