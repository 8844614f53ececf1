"""GPU parity tests proper: the gfx950 kernels, called through the operator API / C ABI, against
(1) the committed golden vectors generated from the reference's dense PyTorch statement and
(2) the float64 CPU oracle on seeded synthetic scenes, plus size-independent properties at the
BASELINE configuration sizes.  Tolerances (fp32 kernels vs float64 truth), stated once:

  forward RGBA            max-abs err <= 2e-4 * max(1, max|rgba|)
  grad_template           max-abs err <= 1e-3 * max|g|        (fp32 atomics, order-nondeterministic sums)
  pose grads (pos/rot/scale)  cosine >= 0.9999 and max-abs err <= 3e-2 * max|g| on white-noise slabs
                          (the reference's own fp32 dense oracle sits at 1-1.8e-2 there)
  raydirs                 raydir 2e-6 abs, tminmax 2e-5 * max(1,|t|)
  AABB                    1e-5 * max(1, |coord|)
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from helpers import SAT_ROUNDOFF, FragileRays, cosine, edge_jump_for, load_krt_400940, npf, scene_rays, to_dev

pytestmark = pytest.mark.gpu

FWD_TOL = 2e-4
GT_TOL = 1e-3
POSE_COS = 0.9999
POSE_TOL = 3e-2


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import ava256_amd
    from ava256_amd import _lib
    import ctypes
    buf = ctypes.create_string_buffer(64)
    _lib.check(_lib.get_lib().mvp_device_arch(0, buf, 64), "mvp_device_arch")
    assert buf.value.decode().startswith("gfx950"), buf.value
    return ava256_amd


BACKWARD_MODES = ["prim", "ray", "cap4"]  # primitive-centric | forced ray-centric fallback | tiny list capacity


def _march(ops, raypos, raydir, stepsize, tminmax, primpos, primrot, primscale, template, fadescale, fadeexp,
           grad_out=None, mode="prim", warp=None):
    """Run forward (+ backward with grad_out) through the public operator. Inputs: numpy float64/32.
    grad_out may be an array or a callable(raysat_numpy) -> array (evaluated after the forward).
    mode selects the backward implementation under test (all must agree with the oracle):
      prim: primitive-centric kernel (LDS accumulation); ray: ray-centric kernel with global atomics for
      everything; cap4: per-primitive list capacity 4, so most primitives overflow into the ray-centric kernel
      while the rest stay primitive-centric (mixed ownership inside one call)."""
    from ava256_amd import _hooks as mm
    diag = torch.zeros(8, dtype=torch.int32, device="cuda")
    mm.set_diag_buffer(diag)
    handoff = mm.patched_handoff(cap=4 if mode == "cap4" else None, ray_centric=(mode == "ray"))
    mm.keep_raysat = True
    t = dict(raypos=to_dev(raypos), raydir=to_dev(raydir), tminmax=to_dev(tminmax), primpos=to_dev(primpos),
             primrot=to_dev(primrot), primscale=to_dev(primscale), template=to_dev(template))
    names = ("primpos", "primrot", "primscale", "template")
    if warp is not None:
        t["warp"] = to_dev(warp)
        names = names + ("warp",)
    for k in names:
        t[k].requires_grad_(grad_out is not None)
    with torch.set_grad_enabled(grad_out is not None), handoff:
        rgba = ops.mvpraymarch(t["raypos"], t["raydir"], float(stepsize), t["tminmax"],
                               (t["primpos"], t["primrot"], t["primscale"]), t["template"], t.get("warp"),
                               algo=1 if warp is not None else 0, fadescale=float(fadescale), fadeexp=float(fadeexp))
    grads = None
    if grad_out is not None:
        if callable(grad_out):
            grad_out = grad_out(npf(mm.last_raysat))
        rgba.backward(to_dev(grad_out))
        grads = {k: npf(t[k].grad) for k in names}
    torch.cuda.synchronize()
    d = mm.read_diag()
    if mm.last_pl_count is not None:  # the hand-off words after the backward (include/mvp_abi.h): flags, and who owned what
        NK = t["primpos"].shape[0] * t["primpos"].shape[1]
        cnt = mm.last_pl_count[:NK].to(torch.int64) & 0xffffffff
        d["handoff_flags"] = int(mm.last_pl_count[NK].item()) & 0xffffffff
        d["prims_two_pass"] = int(((cnt >> 30) & 1).sum().item())        # marked for the two-pass (residual) kernel
        d["prims_handed_over"] = int(((cnt >> 31) & 1).sum().item())     # handed to the ray-centric kernel by the backward
    mm.set_diag_buffer(None)
    mm.keep_raysat = False
    mm.last_raysat = mm.last_pl_count = None
    return npf(rgba), grads, d


def _check_grads(mine, ref, what=""):
    e = np.abs(mine["template"] - ref["template"]).max()
    assert e <= GT_TOL * np.abs(ref["template"]).max(), (what, "template", e, np.abs(ref["template"]).max())
    for k in ("primpos", "primrot", "primscale"):
        c = cosine(mine[k], ref[k])
        e = np.abs(mine[k] - ref[k]).max()
        assert c >= POSE_COS, (what, k, c)
        assert e <= POSE_TOL * np.abs(ref[k]).max(), (what, k, e, np.abs(ref[k]).max())


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", BACKWARD_MODES)
@pytest.mark.parametrize("name", ["march_k8_m8", "march_k64_m4", "march_k8_m8_sat", "march_k125_m4_fade"])
def test_march_matches_reference_golden(ops, name, mode):
    """HIP forward + backward vs the fixtures made from mvpraymarch.py:553-641 (float64)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    rgba, grads, diag = _march(ops, g["raypos"], g["raydir"], g["stepsize"], g["tminmax"], g["primpos"],
                               g["primrot"], g["primscale"], g["template"], g["fadescale"], g["fadeexp"],
                               grad_out=np.ones_like(g["rgba"]), mode=mode)
    assert diag["list_overflow"] == 0 and diag["frontier_overflow"] == 0
    assert np.abs(rgba - g["rgba"]).max() <= FWD_TOL * max(1.0, np.abs(g["rgba"]).max())
    mine = dict(template=grads["template"] * g["chain_template"], primpos=grads["primpos"] * g["chain_primpos"],
                primrot=grads["primrot"], primscale=grads["primscale"] * g["chain_primscale"])
    ref = dict(template=g["graw_template"], primpos=g["graw_primpos"], primrot=g["graw_primrot"],
               primscale=g["graw_primscale"])
    _check_grads(mine, ref, name)


SCENES = [
    # N, H, W, K, alpha_gain, slab
    (2, 64, 64, 512, 1.0, 8),     # unsaturated, shell scene
    (1, 50, 37, 512, 40.0, 8),    # ragged packets (W,H not multiples of 8) + about half the rays saturate
    (2, 40, 40, 37, 8.0, 8),      # K not a power of two: DFS leaf order differs from ascending k
    (1, 33, 65, 1, 30.0, 8),      # a single primitive (root is a leaf)
    (1, 48, 48, 300, 5.0, 4),     # 4^3 slabs, K not a power of two
    (1, 24, 24, 2, 30.0, 8),
    (1, 256, 256, 128, 3.0, 8),   # ~160k samples per primitive: the integer LDS accumulators are drained several times
]


@pytest.mark.parametrize("mode", BACKWARD_MODES)
@pytest.mark.parametrize("cfg", SCENES, ids=lambda c: "N%d_%dx%d_K%d_a%g_s%d" % c)
def test_march_matches_oracle_on_synthetic_scenes(ops, oracle64, cfg, mode):
    from ava256_amd.scene import make_scene
    N, H, W, K, again, slab = cfg
    s = make_scene(N, H, W, K, device="cpu", seed=7 + K, alpha_gain=again, slab=slab)
    if K < 100:  # fewer, bigger boxes
        s["primscale"] = s["primscale"] * 0.5
    rp, rd, tm = scene_rays(oracle64, s)
    a = (rp, rd, s["stepsize"], tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy(),
         s["template"].numpy())
    ref_rgba, ref_sat, st = oracle64.march_forward(*a, ray_diagnostics=True)
    assert st["list_overflow"] == 0 and st["rays_hit"] > 0
    rng = np.random.default_rng(3)
    gout = rng.normal(size=ref_rgba.shape)
    # rays the ORACLE calls borderline may saturate elsewhere in fp32 (helpers.FragileRays); anything else must agree
    fragile = FragileRays(ref_sat, st["margin"], gout, nsamples=st["nsamples"])
    rgba, grads, diag = _march(ops, *a, 8.0, 8.0, grad_out=fragile, mode=mode)
    assert diag["list_overflow"] == 0 and diag["frontier_overflow"] == 0
    assert diag["packets_hit"] > 0
    fr = fragile.mask
    g2 = fragile.masked()
    rgp, rgr, rgs, rgt = oracle64.march_backward(*a, ref_sat, g2)
    scale = max(1.0, np.abs(ref_rgba).max())
    err = np.abs(rgba - ref_rgba).max(-1)
    assert (err[~fr] > FWD_TOL * scale).sum() == 0, (err[~fr].max(), scale)
    assert np.abs(rgba[..., 3] - ref_rgba[..., 3]).max() <= FWD_TOL          # alpha agrees on fragile rays too
    ref = dict(template=rgt, primpos=rgp, primrot=rgr, primscale=rgs)
    _check_grads(grads, ref, str(cfg))


@pytest.mark.parametrize("mode", BACKWARD_MODES)
@pytest.mark.parametrize("shape,fadescale,fadeexp", [((4, 6, 5), 5.0, 6.0), ((8, 8, 8), 7.0, 3.0), ((3, 2, 7), 8.0, 8.0)])
def test_non_cubic_slabs_and_general_fade(ops, oracle64, shape, fadescale, fadeexp, mode):
    """The template's (TD, TH, TW) come from the tensor (mvpraymarch.cpp:233-238) and need not be equal; fadeexp != 8
    takes the pow form of the fade and of its derivative (primsampler.h:48-51,70-74).  Covers the generic-stride
    instantiations of both kernels with either fade, and the 8^3 one with the general fade."""
    from ava256_amd.scene import make_scene
    N, H, W, K = 2, 40, 48, 96
    s = make_scene(N, H, W, K, device="cpu", seed=31, alpha_gain=3.0, slab=4)
    s["primscale"] = s["primscale"] * 0.7
    rng = np.random.default_rng(17)
    TD, TH, TW = shape
    tpl = np.concatenate([np.maximum(100 + 25 * rng.normal(size=(N, K, TD, TH, TW, 3)), 0),
                          3.0 * np.exp(0.1 * rng.normal(size=(N, K, TD, TH, TW, 1)))], axis=-1)
    rp, rd, tm = scene_rays(oracle64, s)
    a = (rp, rd, s["stepsize"], tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy(), tpl)
    ref_rgba, ref_sat, st = oracle64.march_forward(*a, fadescale=fadescale, fadeexp=fadeexp, ray_diagnostics=True)
    assert st["rays_hit"] > 0 and st["list_overflow"] == 0
    gout = rng.normal(size=ref_rgba.shape)
    fragile = FragileRays(ref_sat, st["margin"], gout, nsamples=st["nsamples"])
    rgba, grads, diag = _march(ops, *a, fadescale, fadeexp, grad_out=fragile, mode=mode)
    fr = fragile.mask
    g2 = fragile.masked()
    rgp, rgr, rgs, rgt = oracle64.march_backward(*a, ref_sat, g2, fadescale=fadescale, fadeexp=fadeexp)
    err = np.abs(rgba - ref_rgba).max(-1)
    assert (err[~fr] > FWD_TOL * max(1.0, np.abs(ref_rgba).max())).sum() == 0, err[~fr].max()
    _check_grads(grads, dict(template=rgt, primpos=rgp, primrot=rgr, primscale=rgs), str(shape))


FUZZ_ESCAPES = []   # (seed, which bound) of the draws that needed the fp32-oracle criterion (see the test below them)


def fuzz_draw(seed, oracle64):
    """The seeded random configuration of test_randomized_configurations (also replayed by tools/diag_fuzz_replay.py)."""
    from ava256_amd.scene import make_scene
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(1, 4))
    H, W = int(rng.integers(9, 71)), int(rng.integers(9, 71))
    K = int(rng.choice([1, 2, 3, 7, 8, 33, 64, 100, 257, 512, 700]))
    rng3 = np.random.default_rng(9000 + seed)
    if seed >= 10 and rng3.random() < 0.2:
        # few primitives under many packets: a primitive's list runs to hundreds of entries and tens of thousands of
        # samples, i.e. several accumulation rounds per primitive, and lists beyond the first capacity guess
        N, H, W = int(rng3.integers(1, 3)), int(rng3.integers(90, 181)), int(rng3.integers(90, 181))
        K = int(rng3.choice([16, 33, 64, 100]))
    shape = tuple(int(x) for x in rng.integers(2, 10, size=3)) if rng.random() < 0.5 else (8, 8, 8)
    again = float(rng.choice([0.5, 2.0, 8.0, 30.0]))
    fadescale, fadeexp = (8.0, 8.0) if rng.random() < 0.5 else (float(rng.uniform(3, 9)), float(rng.uniform(2.5, 9)))
    mode = str(rng.choice(BACKWARD_MODES))
    s = make_scene(N, H, W, K, device="cpu", seed=50 + seed, alpha_gain=1.0, slab=4)
    s["primscale"] = s["primscale"] * float(rng.uniform(0.35, 1.0) if K < 100 else rng.uniform(0.7, 1.2))
    stepsize = float(s["stepsize"]) * float(rng.choice([0.5, 1.0, 2.0]))
    TD, TH, TW = shape
    tpl = np.concatenate([np.maximum(100 + 25 * rng.normal(size=(N, K, TD, TH, TW, 3)), 0),
                          again * np.exp(0.1 * rng.normal(size=(N, K, TD, TH, TW, 1)))], axis=-1)
    warp = None
    if rng.random() < 0.25:  # identity grid + noise, WD x WH x WW nodes
        WD, WH, WW = (int(x) for x in rng.integers(2, 9, size=3))
        zz, yy, xx = np.meshgrid(np.linspace(-1, 1, WD), np.linspace(-1, 1, WH), np.linspace(-1, 1, WW), indexing="ij")
        warp = np.stack([xx, yy, zz], -1)[None, None] + float(rng.choice([0.05, 0.3])) * rng.normal(size=(N, K, WD, WH, WW, 3))
    rp, rd, tm = scene_rays(oracle64, s)
    a = (rp, rd, stepsize, tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy(), tpl)
    gout = rng.normal(size=(N, H, W, 4))
    gstyle = "normal"
    if seed >= 10:  # (the first ten draws are the default run and stay what they were) shape of the upstream gradient
        rng2 = np.random.default_rng(7000 + seed)
        gstyle = str(rng2.choice(["normal", "lognormal", "outliers", "sparse", "rgb-only"]))
        if gstyle == "lognormal":      # per-ray magnitudes over ~5 decades
            gout *= np.exp(3.0 * rng2.normal(size=gout.shape[:3]))[..., None]
        elif gstyle == "outliers":     # a few pixels 1e4 times the rest
            gout[rng2.random(size=gout.shape[:3]) < 3e-3] *= 1.0e4
        elif gstyle == "sparse":       # most rays carry no gradient at all (masked loss)
            gout[rng2.random(size=gout.shape[:3]) < 0.9] = 0.0
        elif gstyle == "rgb-only":
            gout[..., 3] = 0.0
    cfg = "seed %d: N%d %dx%d K%d slab%s gain%g fade(%g,%g) dt%g mode %s warp %s grad %s" % (
        seed, N, H, W, K, shape, again, fadescale, fadeexp, stepsize, mode, None if warp is None else warp.shape[2:5], gstyle)
    return dict(N=N, K=K, args=a, fadescale=fadescale, fadeexp=fadeexp, mode=mode, warp=warp, gout=gout, gstyle=gstyle, cfg=cfg)


@pytest.mark.parametrize("seed", [int(os.environ.get("MVP_FUZZ_FIRST", "0")) + i  # more draws: MVP_FUZZ_SEEDS=n, from
                                  for i in range(int(os.environ.get("MVP_FUZZ_SEEDS", "24")))])  # seed MVP_FUZZ_FIRST on
def test_randomized_configurations(ops, oracle64, oracle32, seed):
    """Seeded random draws over image size (ragged packets), primitive count (non powers of two, tiny), slab shape,
    opacity (none to most rays saturating), box size, step size and fade parameters, and -- one draw in four -- a warp
    field (algo 1) on a random grid; forward and all gradients against the float64 oracle with the standing tolerances,
    backward owner chosen at random as well.  Seeds >= 10 also draw the shape of the upstream gradient."""
    c = fuzz_draw(seed, oracle64)
    N, K, a, fadescale, fadeexp, mode, warp, gout, gstyle, cfg = (c[k] for k in (
        "N", "K", "args", "fadescale", "fadeexp", "mode", "warp", "gout", "gstyle", "cfg"))
    ref_rgba, ref_sat, st = oracle64.march_forward(*a, fadescale=fadescale, fadeexp=fadeexp, ray_diagnostics=True, warp=warp)
    if st["rays_hit"] == 0 or st["list_overflow"] > 0:
        pytest.skip("degenerate draw")
    fragile = FragileRays(ref_sat, st["margin"], gout, nsamples=st["nsamples"], max_frac=0.01, min_allowed=3, edge=st["edge"],
                          edge_jump=edge_jump_for(FWD_TOL * max(1.0, np.abs(ref_rgba).max()), a[7]), sat_roundoff=SAT_ROUNDOFF)
    rgba, grads, diag = _march(ops, *a, fadescale, fadeexp, grad_out=fragile, mode=mode, warp=warp)
    fr = fragile.mask
    g2 = fragile.masked()
    ref = oracle64.march_backward(*a, ref_sat, g2, fadescale=fadescale, fadeexp=fadeexp, warp=warp)
    rgp, rgr, rgs, rgt = ref[:4]
    err = np.abs(rgba - ref_rgba).max(-1)
    assert (err[~fr] > FWD_TOL * max(1.0, np.abs(ref_rgba).max())).sum() == 0, (cfg, err[~fr].max())
    if gstyle in ("lognormal", "outliers"):
        # Magnitudes spread over decades: the global max-abs bounds of _check_grads say nothing about the primitives the
        # large rays miss.  Every primitive's slab gradient against ITS OWN max |g| instead -- plus the absolute floor of
        # fixed-point accumulation: a quantum is <= 2^14 / 2^31 of the a-priori bound of a round's values,
        # B_rgb = G min(1, Amax_k dt), B_a = G dt (3 (Tmax_k + Rmax) + 1) with G the largest upstream gradient.  (A box
        # that these tiny images only graze at a corner has fade ~ e^-20 on every sample: its whole gradient sits below
        # the quantum.  The fp32-atomic kernel is held to the same bound.)
        tplk = a[7].reshape(N * K, -1, 4)
        G, dt = np.abs(g2).max(), float(a[2])
        Brgb = G * np.minimum(1.0, np.abs(tplk[..., 3]).max(1) * dt)
        Ba = G * dt * (3.0 * (np.abs(tplk[..., :3]).max((1, 2)) + np.abs(tplk[..., :3]).max()) + 1.0)
        e = np.abs(grads["template"].reshape(N * K, -1, 4) - rgt.reshape(N * K, -1, 4))
        pmax = np.abs(rgt.reshape(N * K, -1, 4))
        e32 = None
        for name, ek, pk, Bk, ch in (("rgb", e[..., :3].max((1, 2)), pmax[..., :3].max((1, 2)), Brgb, slice(0, 3)),
                                     ("alpha", e[..., 3].max(1), pmax[..., 3].max(1), Ba, slice(3, 4))):
            over = ek - (GT_TOL * pk + 1e-5 * Bk)
            if over.max() > 0:
                # as below: acceptable only where plain fp32 (the oracle's f32 build, same raysat and gradients) is
                # further from float64 on that very primitive (huge boxes: hundreds of samples per ray and primitive)
                FUZZ_ESCAPES.append((seed, "per-primitive " + name))
                print("fuzz escape hatch taken:", cfg, name)
                if e32 is None:
                    r32 = oracle32.march_backward(*a, ref_sat, g2, fadescale=fadescale, fadeexp=fadeexp, warp=warp)
                    e32 = np.abs(r32[3].reshape(N * K, -1, 4) - rgt.reshape(N * K, -1, 4))
                over = ek - np.maximum(GT_TOL * pk + 1e-5 * Bk, 1.5 * e32[..., ch].max((1, 2)))
            assert over.max() <= 0, (cfg, "per-primitive template " + name, int(over.argmax()), float(ek[over.argmax()]),
                                     float(pk[over.argmax()]), float(Bk[over.argmax()]))
        for k, refg in (("primpos", rgp), ("primrot", rgr), ("primscale", rgs)):
            assert cosine(grads[k], refg) >= POSE_COS, (cfg, k, cosine(grads[k], refg))
    else:
        try:
            _check_grads(grads, dict(template=rgt, primpos=rgp, primrot=rgr, primscale=rgs), cfg)
        except AssertionError:
            FUZZ_ESCAPES.append((seed, "standing bounds"))
            print("fuzz escape hatch taken:", cfg)
            # Over the standing bound: acceptable only where fp32 itself is the limit (a few huge boxes, hundreds of
            # samples per ray and primitive) -- the same march in plain fp32 (the reference's arithmetic; the oracle's
            # f32 build, same raysat and gradients) must then be further from float64 than the kernel is.
            r32 = oracle32.march_backward(*a, ref_sat, g2, fadescale=fadescale, fadeexp=fadeexp, warp=warp)
            for k, mine, r64, f32 in (("template", grads["template"], rgt, r32[3]), ("primpos", grads["primpos"], rgp, r32[0]),
                                      ("primrot", grads["primrot"], rgr, r32[1]), ("primscale", grads["primscale"], rgs, r32[2])):
                e_k, e_32 = np.abs(mine - r64).max(), np.abs(f32 - r64).max()
                tol = (GT_TOL if k == "template" else POSE_TOL) * np.abs(r64).max()
                assert e_k <= max(tol, 1.5 * e_32), (cfg, k, "kernel", e_k, "fp32 oracle", e_32, "bound", tol)
    if warp is not None:  # a position gradient like the pose gradients (see the warp-field tests below for the bounds)
        gw, rgw = grads["warp"], ref[4]
        assert cosine(gw, rgw) >= POSE_COS and np.linalg.norm(gw - rgw) <= 2e-2 * np.linalg.norm(rgw), (cfg, cosine(gw, rgw))


def test_fuzz_escape_hatch_is_not_taken_on_the_default_seeds():
    """test_randomized_configurations lets a draw exceed the standing bounds when the oracle's own fp32 build is 1.5 x worse
    still -- honest (fp32 is then the limit), but an escape hatch: it is counted, printed, and on the default 24 seeds it
    must not be taken at all.  (Runs after the draws: same module, file order.  With MVP_FUZZ_SEEDS / MVP_FUZZ_FIRST set
    the count is only reported: one draw in ~500 takes it)"""
    print("fuzz escape hatch taken %d times: %s" % (len(FUZZ_ESCAPES), FUZZ_ESCAPES))
    if "MVP_FUZZ_SEEDS" not in os.environ and "MVP_FUZZ_FIRST" not in os.environ:
        assert FUZZ_ESCAPES == [], FUZZ_ESCAPES


@pytest.mark.parametrize("mode", BACKWARD_MODES)
@pytest.mark.parametrize("name", ["march_warp_k8_m8", "march_warp_k8_m8_sat"])
def test_warp_sampler_matches_reference_golden(ops, name, mode):
    """algo 1 (PrimSamplerTW<true>): fixtures from the reference's gradcheck(dowarp=True) dense loop (float64).
    Backward by the primitive-centric kernel's warp-field variant, the ray-centric kernel, and both (cap4)."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    rgba, grads, diag = _march(ops, g["raypos"], g["raydir"], g["stepsize"], g["tminmax"], g["primpos"],
                               g["primrot"], g["primscale"], g["template"], g["fadescale"], g["fadeexp"],
                               grad_out=np.ones_like(g["rgba"]), warp=g["warp"], mode=mode)
    assert np.abs(rgba - g["rgba"]).max() <= FWD_TOL * max(1.0, np.abs(g["rgba"]).max())
    mine = dict(template=grads["template"] * g["chain_template"], primpos=grads["primpos"] * g["chain_primpos"],
                primrot=grads["primrot"], primscale=grads["primscale"] * g["chain_primscale"])
    ref = dict(template=g["graw_template"], primpos=g["graw_primpos"], primrot=g["graw_primrot"],
               primscale=g["graw_primscale"])
    _check_grads(mine, ref, name)
    gw, rw = grads["warp"], g["graw_warp"]
    assert cosine(gw, rw) >= POSE_COS and np.abs(gw - rw).max() <= POSE_TOL * np.abs(rw).max()


@pytest.mark.parametrize("mode", BACKWARD_MODES)
@pytest.mark.parametrize("wshape,noise,fadescale,fadeexp,gw_maxabs", [((4, 4, 4), 0.15, 8.0, 8.0, 1e-1),
                                                                     ((3, 5, 2), 0.6, 6.0, 5.0, 1e-1),
                                                                     ((8, 8, 8), 0.3, 8.0, 8.0, 2e-1)],
                         ids=["w4_n0.15", "w3x5x2_n0.6_fade5", "w8_n0.3"])
def test_warp_sampler_matches_oracle_on_a_shell_scene(ops, oracle64, wshape, noise, fadescale, fadeexp, gw_maxabs, mode):
    """Warp field = identity grid + noise, ragged image, K not a power of two.  Grids: cubic 4^3, non-cubic 3x5x2 with the
    general fade and a warp strong enough that warped coordinates leave the slab (zero padding, utils.h:475-498), and the
    8^3 grid of the reference's gradcheck.  Every owner of the backward (primitive-centric warp variant, ray-centric
    kernel, mixed) must match the float64 oracle, grad_warp included."""
    from ava256_amd.scene import make_scene
    N, H, W, K = 2, 45, 52, 300
    s = make_scene(N, H, W, K, device="cpu", seed=21, alpha_gain=4.0)
    g = torch.Generator().manual_seed(5)
    WD, WH, WW = wshape
    zz, yy, xx = torch.meshgrid(torch.linspace(-1.0, 1.0, WD), torch.linspace(-1.0, 1.0, WH), torch.linspace(-1.0, 1.0, WW),
                                indexing="ij")
    ident = torch.stack([xx, yy, zz], dim=-1)                                  # warp[z,y,x] = (x,y,z): identity
    warp = (ident[None, None] + noise * torch.randn(N, K, WD, WH, WW, 3, generator=g)).contiguous().numpy()
    rp, rd, tm = scene_rays(oracle64, s)
    a = (rp, rd, s["stepsize"], tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy(),
         s["template"].numpy())
    ref_rgba, ref_sat, st = oracle64.march_forward(*a, warp=warp, fadescale=fadescale, fadeexp=fadeexp, ray_diagnostics=True)
    gout = np.random.default_rng(8).normal(size=ref_rgba.shape)
    fragile = FragileRays(ref_sat, st["margin"], gout, nsamples=st["nsamples"])
    rgba, grads, diag = _march(ops, *a, fadescale, fadeexp, grad_out=fragile, warp=warp, mode=mode)
    fr = fragile.mask
    g2 = fragile.masked()
    rgp, rgr, rgs, rgt, rgw = oracle64.march_backward(*a, ref_sat, g2, warp=warp, fadescale=fadescale, fadeexp=fadeexp)
    err = np.abs(rgba - ref_rgba).max(-1)
    assert (err[~fr] > FWD_TOL * max(1.0, np.abs(ref_rgba).max())).sum() == 0
    _check_grads(grads, dict(template=rgt, primpos=rgp, primrot=rgr, primscale=rgs), "warp scene")
    # grad_warp is a scatter of dL/dy1 (a position gradient: white-noise slabs make it cancel heavily, like the pose
    # gradients); held to cosine >= 0.9999, norm-wise 1e-2 and max-abs 1e-1 (2e-1 on the 8^3 grid, whose nodes collect
    # fewer samples each).  Calibration: the float32 build of the ORACLE against its float64 build on these three
    # scenes gives cosine 0.999994 / 0.999992 / 0.999978, norm-wise 3.5e-3 / 4.1e-3 / 6.6e-3 and max-abs
    # 3.5e-2 / 1.4e-2 / 9.7e-2 -- the kernels sit in the same place (all three backward owners agree to 1e-6).
    gw = grads["warp"]
    stats = (cosine(gw, rgw), np.linalg.norm(gw - rgw) / np.linalg.norm(rgw), np.abs(gw - rgw).max() / np.abs(rgw).max())
    assert stats[0] >= POSE_COS and stats[1] <= 1e-2 and stats[2] <= gw_maxabs, stats


def test_heavy_scene_takes_the_exact_traversal_fallback(ops, oracle64):
    """Primitives in raw Fibonacci-spiral order (consecutive k are a golden angle apart): the fixed-order heap has
    no locality, the BFS frontier exceeds its 512-entry capacity and packets fall back to the reference-style DFS.
    Hit lists stay far below the 512 cap, so the result must still match the oracle."""
    from ava256_amd.scene import make_scene
    s = make_scene(1, 32, 32, 4096, device="cpu", seed=99, alpha_gain=3.0, order="fibonacci")
    rp, rd, tm = scene_rays(oracle64, s)
    a = (rp, rd, s["stepsize"], tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy(),
         s["template"].numpy())
    ref_rgba, ref_sat, st = oracle64.march_forward(*a)
    assert st["list_overflow"] == 0
    gout = np.random.default_rng(4).normal(size=ref_rgba.shape)
    ref = dict(zip(("primpos", "primrot", "primscale", "template"), oracle64.march_backward(*a, ref_sat, gout)))
    rgba, grads, diag = _march(ops, *a, 8.0, 8.0, grad_out=gout)
    assert diag["frontier_overflow"] > 0 and diag["list_overflow"] == 0, diag
    assert np.abs(rgba - ref_rgba).max() <= FWD_TOL * max(1.0, np.abs(ref_rgba).max())
    _check_grads(grads, ref, "heavy")


def test_no_grad_mode_and_empty_rays(ops):
    """no-grad forward (raysat=None, mvpraymarch.py:147-152) and rays that miss the volume give zeros."""
    from ava256_amd.scene import make_scene
    s = make_scene(1, 32, 32, 64, device="cuda", seed=5)
    raypos, raydir, tminmax = ops.compute_raydirs(s["campos"], s["camrot"], s["focal"], s["princpt"],
                                                  s["pixelcoords"], s["volradius"])
    with torch.no_grad():
        a = ops.mvpraymarch(raypos, raydir, s["stepsize"], tminmax, (s["primpos"], s["primrot"], s["primscale"]),
                            s["template"], None)
    b = ops.mvpraymarch(raypos, raydir, s["stepsize"], tminmax, (s["primpos"], s["primrot"], s["primscale"]),
                        s["template"], None)
    assert torch.equal(a, b)
    # point every ray away from the volume: tmin > tmax -> zeros
    raypos2 = raypos + 10.0
    _, _, tm2 = None, None, torch.stack([torch.full_like(tminmax[..., 0], 5.0), torch.full_like(tminmax[..., 0], 1.0)], -1)
    z = ops.mvpraymarch(raypos2, raydir, s["stepsize"], tm2.contiguous(), (s["primpos"], s["primrot"], s["primscale"]),
                        s["template"], None)
    assert torch.count_nonzero(z) == 0


def test_degenerate_shapes(ops):
    """Empty batch, single-pixel image, zero-height image: shapes come back right and nothing is launched out of
    bounds (the reference's launchers would be handed grid dimension 0)."""
    from ava256_amd.scene import make_scene
    s = make_scene(2, 8, 8, 16, device="cuda", seed=2)
    prim = (s["primpos"], s["primrot"], s["primscale"])
    for n, h, w in [(0, 8, 8), (2, 1, 1), (2, 0, 8), (2, 3, 0)]:
        sl = slice(0, n)
        pc = s["pixelcoords"][sl, :h, :w].contiguous()
        rp, rd, tm = ops.compute_raydirs(s["campos"][sl], s["camrot"][sl], s["focal"][sl], s["princpt"][sl], pc, 256.0)
        assert rp.shape == (n, h, w, 3) and tm.shape == (n, h, w, 2)
        t = s["template"][sl].clone().requires_grad_(True)
        out = ops.mvpraymarch(rp, rd, s["stepsize"], tm, tuple(x[sl] for x in prim), t, None)
        assert out.shape == (n, h, w, 4)
        out.sum().backward()
        assert t.grad.shape == t.shape and torch.isfinite(t.grad).all()
        if n * h * w == 0:
            assert float(t.grad.abs().sum()) == 0.0


def test_very_fine_steps_use_the_unpacked_path(ops, oracle64):
    """stepsize so small that lattice-step indices exceed the packed 16-bit ranges / 23-bit sample keys: the forward
    falls back to unranged sweeping and raises the global flag, so the backward must come from the ray-centric kernel."""
    from ava256_amd.scene import make_scene
    s = make_scene(1, 8, 8, 4, device="cpu", seed=12, alpha_gain=0.01)   # faint: nothing saturates early
    s["primscale"] = s["primscale"] * 0.2
    dt = 2.0e-5                                   # ~1e5 steps across the volume
    rp, rd, tm = scene_rays(oracle64, s)
    a = (rp, rd, dt, tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy(), s["template"].numpy())
    ref, ref_sat, st = oracle64.march_forward(*a, ray_diagnostics=True)
    assert st["steps"] / max(1, st["rays_hit"]) > 65535     # more lattice steps per ray than the packed ranges hold
    gout = np.random.default_rng(2).normal(size=ref.shape)
    fragile = FragileRays(ref_sat, st["margin"], gout, nsamples=st["nsamples"])
    rgba, grads, diag = _march(ops, *a, 8.0, 8.0, grad_out=fragile)
    g2 = fragile.masked()
    rg = dict(zip(("primpos", "primrot", "primscale", "template"), oracle64.march_backward(*a, ref_sat, g2)))
    # 1e5 accumulated fp32 samples per ray: forward tolerance scaled by sqrt(steps / 150)
    assert np.abs(rgba - ref)[~fragile.mask].max() <= 30 * FWD_TOL * max(1.0, np.abs(ref).max())
    assert np.abs(grads["template"] - rg["template"]).max() <= 3e-2 * np.abs(rg["template"]).max()
    for k in ("primpos", "primrot", "primscale"):
        assert cosine(grads[k], rg[k]) >= 0.999, k


def test_raydirs_matches_golden_and_oracle(ops, oracle64):
    g = np.load(os.path.join(GOLDEN, "raydirs_small.npz"))
    out = ops.compute_raydirs(to_dev(g["viewpos"]), to_dev(g["viewrot"]), to_dev(g["focal"]), to_dev(g["princpt"]),
                              to_dev(g["pixelcoords"]), float(g["volradius"]))
    raypos, raydir, tminmax = [npf(x) for x in out]
    assert np.abs(raypos - g["raypos"]).max() <= 1e-6
    assert np.abs(raydir - g["raydir"]).max() <= 2e-6
    assert np.abs(tminmax - g["tminmax"]).max() <= 2e-5 * max(1.0, np.abs(g["tminmax"]).max())
    # the reference's own fixture camera (tests/test_extensions.py:44-66), at a reduced, ragged size, both as a
    # pixelcoords tensor and as the (W, H) tuple form (extensions/utils/utils.py:28-33)
    campos, camrot, focal, princpt = load_krt_400940()
    W, H = 333, 517
    px, py = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    pc = np.stack((px, py), -1)[None]
    ref = oracle64.raydirs(campos, camrot, focal, princpt, pc, 256.0)
    for form in (to_dev(pc), (W, H)):
        out = ops.compute_raydirs(to_dev(campos), to_dev(camrot), to_dev(focal), to_dev(princpt), form, 256.0)
        assert out[0].shape == (1, H, W, 3) and out[1].shape == (1, H, W, 3) and out[2].shape == (1, H, W, 2)
        assert np.abs(npf(out[0]) - ref[0]).max() <= 1e-5
        assert np.abs(npf(out[1]) - ref[1]).max() <= 2e-6
        t = ref[2]
        assert np.abs(npf(out[2]) - t).max() <= 2e-5 * max(1.0, np.abs(t).max())


@pytest.mark.parametrize("K", [1, 2, 3, 37, 128, 129, 300, 4096, 5000])
def test_aabb_matches_oracle(ops, oracle64, K):
    from ava256_amd.mvpraymarch import build_accel
    from ava256_amd.scene import make_primitives
    p = make_primitives(2, K, device="cpu", seed=K)
    ref = oracle64.aabb(p["primpos"].numpy(), p["primrot"].numpy(), p["primscale"].numpy())
    _, _, A = build_accel((p["primpos"].cuda(), p["primrot"].cuda(), p["primscale"].cuda()), 0, fixedorder=True)
    A = npf(A)
    assert A.shape == ref.shape
    assert np.abs(A - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def test_reference_extension_shape_test(ops):
    """Counterpart of the reference's tests/test_extensions.py:69-103 (same camera, image size, K, random
    primitives): only shapes are asserted there.  Its random primscale in (0,1) makes every box cover the
    whole volume (a degenerate scene), but see the note at the end: no ray reaches the volume at all."""
    from ava256_amd import _hooks as mm
    campos, camrot, focal, princpt = [to_dev(x) for x in load_krt_400940()]
    imwidth, imheight = 1334, 2048
    px, py = np.meshgrid(np.arange(imwidth, dtype=np.float32), np.arange(imheight, dtype=np.float32))
    pixelcoords = torch.from_numpy(np.stack((px, py), axis=-1))[None].cuda()
    raypos, raydir, tminmax = ops.compute_raydirs(campos, camrot, focal, princpt, pixelcoords, 256.0)
    assert raypos.shape == (1, imheight, imwidth, 3) and tminmax.shape == (1, imheight, imwidth, 2)
    torch.manual_seed(0)
    K = 128 ** 2
    decout = {
        "template": torch.rand(1, K, 8, 8, 8, 4).cuda(),
        "primpos": torch.rand(1, K, 3).cuda(),
        "primrot": torch.rand(1, K, 3, 3).cuda(),
        "primscale": torch.rand(1, K, 3).cuda(),
    }
    diag = torch.zeros(8, dtype=torch.int32, device="cuda")
    mm.set_diag_buffer(diag)
    with torch.no_grad():
        rayrgb, rayalpha, rayrgba, pos_img = ops.Raymarcher(256.0)(raypos, raydir, tminmax, decout)
    torch.cuda.synchronize()
    d = mm.read_diag()
    mm.set_diag_buffer(None)
    assert rayrgb.shape == (1, 3, imheight, imwidth)
    assert rayalpha.shape == (1, 1, imheight, imwidth)
    assert rayrgba.shape == (1, 4, imheight, imwidth)
    assert pos_img is None
    assert torch.isfinite(rayrgb).all()
    # with this camera and half-size image every ray misses the [-1,1]^3 volume (tmin > tmax), so the
    # reference's own test renders nothing; all packets must exit before touching a primitive
    assert d["packets_hit"] == 0 and float(rayalpha.abs().max()) == 0.0


def test_drop_in_boundary_matches_reference_glue(ops):
    """The drop-in claim itself (oracle O2).  tests/golden/boundary_raymarcher.npz holds what the
    REFERENCE'S OWN Python glue (compute_raydirs, mvpraymarch/MVPRaymarch/build_accel, Raymarcher -- imported unmodified
    from the reference tree) returns when its two native modules are float64 stand-ins with the reference's positional
    signatures: rays, (rayrgb, rayalpha) in NCHW, and the gradients of the decoder outputs for a weighted-sum loss,
    with renderoptions that carry an option of mvpraymarch (fadescale=6) and a key it does not know.  Here the same
    inputs go through THIS build, imported through the reference's import paths."""
    from extensions.utils.utils import compute_raydirs          # models/autoencoder.py:19
    from models.raymarchers.mvpraymarcher import Raymarcher     # models/autoencoder.py:20
    from ava256_amd import _hooks as mm
    g = np.load(os.path.join(GOLDEN, "boundary_raymarcher.npz"))
    assert list(g["calls"])[:2] == ["compute_raydirs_forward", "compute_aabb"]  # the call order the glue produced
    cam = [to_dev(g["in_" + k]) for k in ("campos", "camrot", "focal", "princpt", "pixelcoords")]
    volradius = float(g["volradius"])
    raypos, raydir, tminmax = compute_raydirs(*cam, volradius)
    for name, t in (("raypos", raypos), ("raydir", raydir), ("tminmax", tminmax)):
        assert np.abs(npf(t) - g[name]).max() <= 2e-6 * max(1.0, np.abs(g[name]).max()), name
    decout = {k: to_dev(g["in_" + k]).requires_grad_(True) for k in ("primpos", "primrot", "primscale", "template")}
    renderoptions = {"fadescale": 6.0, "fadeexp": 8.0, "not_an_option_of_mvpraymarch": 123}
    assert sorted(renderoptions) == list(g["renderoptions_keys"])
    mm.keep_raysat = True
    rm = Raymarcher(volradius, dt=float(g["dt"]))
    assert len(list(rm.parameters())) == 0 and len(list(rm.buffers())) == 0  # checkpoints load unchanged
    rayrgb, rayalpha, rayrgba, pos_img = rm(raypos, raydir, tminmax, decout, renderoptions=renderoptions)
    assert pos_img is None and tuple(rayrgba.shape) == tuple(g["rayrgba_view_shape"])
    assert rayrgb.is_contiguous() and rayalpha.is_contiguous()
    tol = 2e-4 * max(1.0, np.abs(g["rayrgb"]).max())
    assert np.abs(npf(rayrgb) - g["rayrgb"]).max() <= tol
    assert np.abs(npf(rayalpha) - g["rayalpha"]).max() <= 2e-4
    # gradients: rays whose saturation state differs from the float64 run by fp32 round-off are left out of the loss
    sat_ref = g["rayalpha"][:, 0] >= 1.0 - 1e-12
    sat_here = npf(mm.last_raysat)[..., 0] > -1.0
    agree = torch.from_numpy((sat_ref == sat_here)[:, None].astype(np.float32)).cuda()
    assert float(agree.mean()) > 0.97
    w_rgb, w_a = to_dev(g["w_rgb"]), to_dev(g["w_a"])
    if float(agree.min()) == 1.0:
        loss = (rayrgb * w_rgb).sum() + (rayalpha * w_a).sum()
        loss.backward()
        assert abs(float(loss) - float(g["loss"])) <= 2e-4 * max(1.0, abs(float(g["loss"])))
        for k in ("template", "primpos", "primrot", "primscale"):
            got, ref = npf(decout[k].grad), g["grad_" + k]
            if k == "template":
                assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max(), k
            else:
                assert cosine(got, ref) >= 0.9999, k
                assert np.abs(got - ref).max() <= 3e-2 * np.abs(ref).max(), k
    else:
        # (not expected for this fixture) some ray saturates in one arithmetic and not in the other: the fixture's gradients
        # belong to the unmasked loss, so the comparison runs on the agreeing rays against the float64 oracle marched on the
        # fixture's own rays with the same options -- a masked loss on both sides, never a skip
        from oracle.mvp_oracle import Oracle
        o64 = Oracle("f64")
        a = (g["raypos"], g["raydir"], float(g["dt"]) / volradius, g["tminmax"], g["in_primpos"], g["in_primrot"],
             g["in_primscale"], g["in_template"])
        ref_rgba, ref_sat, _ = o64.march_forward(*a, fadescale=6.0, fadeexp=8.0)
        assert np.abs(ref_rgba[..., :3].transpose(0, 3, 1, 2) - g["rayrgb"]).max() <= 1e-9 * max(1.0, np.abs(g["rayrgb"]).max())
        gout = np.ascontiguousarray(np.concatenate([g["w_rgb"], g["w_a"]], axis=1).transpose(0, 2, 3, 1) * (sat_ref == sat_here)[..., None])
        rgp, rgr, rgs, rgt = o64.march_backward(*a, ref_sat, gout, fadescale=6.0, fadeexp=8.0)
        loss = ((rayrgb * w_rgb).sum(1, keepdim=True) * agree).sum() + (rayalpha * w_a * agree).sum()
        loss.backward()
        for k, ref in (("template", rgt), ("primpos", rgp), ("primrot", rgr), ("primscale", rgs)):
            got = npf(decout[k].grad)
            if k == "template":
                assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max(), k
            else:
                assert cosine(got, ref) >= 0.9999, k
                assert np.abs(got - ref).max() <= 3e-2 * np.abs(ref).max(), k
    mm.keep_raysat = False


@pytest.mark.parametrize("bad", [float("nan"), float("inf")])
def test_non_finite_upstream_gradient(ops, oracle64, bad):
    """A NaN / Inf in grad_rayrgba (the training loop zeroes such gradients AFTER backward, ddp-train.py:436-439): the
    fixed-point path cannot bound its sums, so every primitive is handed to the ray-centric kernel, whose fp32 atomics
    propagate the value exactly where the reference's would -- NaN in the gradients of everything the poisoned ray
    touches, finite and correct elsewhere."""
    from ava256_amd.scene import make_scene
    s = make_scene(1, 32, 32, 64, device="cpu", seed=4, alpha_gain=0.02, slab=8)  # thin: no ray saturates
    s["primscale"] = s["primscale"] * 0.6
    rp, rd, tm = scene_rays(oracle64, s)
    a = (rp, rd, s["stepsize"], tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy(),
         s["template"].numpy())
    ref_rgba, ref_sat, st = oracle64.march_forward(*a)
    hit = np.argwhere(ref_rgba[0, :, :, 3] > 0)
    assert len(hit) > 10 and st["rays_saturated"] == 0
    rng = np.random.default_rng(8)
    gout = rng.normal(size=ref_rgba.shape)
    y, x = hit[len(hit) // 2]
    gout[0, y, x, 1] = bad
    rgba, grads, diag = _march(ops, *a, 8.0, 8.0, grad_out=gout, mode="prim")
    with np.errstate(invalid="ignore", over="ignore"):
        rgp, rgr, rgs, rgt = oracle64.march_backward(*a, ref_sat, gout)
    for k, ref in (("template", rgt), ("primpos", rgp), ("primrot", rgr), ("primscale", rgs)):
        got = grads[k]
        badmask = ~np.isfinite(ref)
        assert badmask.any() and not badmask.all(), k
        assert (~np.isfinite(got[badmask])).all(), k                # poisoned where the reference semantics poison
        ok = ~badmask
        assert np.isfinite(got[ok]).all(), k
        tol = (1e-3 if k == "template" else 3e-2) * np.abs(ref[ok]).max()
        assert np.abs(got[ok] - ref[ok]).max() <= tol, k


def test_rgba_split_is_bit_exact(ops):
    """Raymarcher's NHWC -> (rgb, alpha) NCHW split (mvpraymarcher.py:50-51) in one pass each way: pure data movement,
    so forward and backward equal the eager permute / slice / contiguous exactly -- also when only one of the two
    outputs feeds the loss, and when the third return value (the permuted view) is used as well."""
    from ava256_amd.raymarcher import split_rgba_nchw
    torch.manual_seed(3)
    for shape in [(2, 37, 50, 4), (1, 1, 1, 4), (3, 8, 8, 4)]:
        x = torch.randn(*shape, device="cuda", requires_grad=True)
        xr = x.detach().clone().requires_grad_(True)
        rgb, alpha, view = split_rgba_nchw(x)
        e = xr.permute(0, 3, 1, 2)
        ergb, ealpha = e[:, :3].contiguous(), e[:, 3:4].contiguous()
        assert rgb.is_contiguous() and alpha.is_contiguous() and view.shape == e.shape
        assert torch.equal(rgb, ergb) and torch.equal(alpha, ealpha) and torch.equal(view, e)
        w1, w2, w3 = torch.randn_like(rgb), torch.randn_like(alpha), torch.randn_like(view)
        ((rgb * w1).sum() + (alpha * w2).sum() + (view * w3).sum()).backward()
        ((ergb * w1).sum() + (ealpha * w2).sum() + (e * w3).sum()).backward()
        assert torch.equal(x.grad, xr.grad)
        x.grad = None
        xr.grad = None
        rgb, alpha, _ = split_rgba_nchw(x)
        (rgb * w1).sum().backward()                      # alpha unused: its incoming gradient is None
        (xr.permute(0, 3, 1, 2)[:, :3].contiguous() * w1).sum().backward()
        assert torch.equal(x.grad, xr.grad)


def test_operator_errors(ops):
    """Error behaviour at the boundary: CPU tensors, wrong dtype, non-contiguous input, unsupported options."""
    from ava256_amd.scene import make_scene
    s = make_scene(1, 16, 16, 8, device="cuda", seed=1)
    rp, rd, tm = ops.compute_raydirs(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"], 256.0)
    prim = (s["primpos"], s["primrot"], s["primscale"])
    with pytest.raises(RuntimeError):
        ops.mvpraymarch(rp.cpu(), rd, s["stepsize"], tm, prim, s["template"], None)
    with pytest.raises(RuntimeError):
        ops.mvpraymarch(rp.double(), rd, s["stepsize"], tm, prim, s["template"], None)
    with pytest.raises(RuntimeError):
        ops.mvpraymarch(rp, rd, s["stepsize"], tm, prim, s["template"].permute(0, 1, 2, 3, 5, 4), None)
    with pytest.raises(NotImplementedError):
        ops.mvpraymarch(rp, rd, s["stepsize"], tm, prim, s["template"], None, usebvh=True)
    with pytest.raises(RuntimeError):
        ops.mvpraymarch(rp, rd, s["stepsize"], tm, prim, s["template"], None, algo=1)      # algo 1 needs a warp field
    with pytest.raises(NotImplementedError):
        ops.mvpraymarch(rp, rd, s["stepsize"], tm, prim, s["template"], None, algo=2)
    # packed [N,K,5,3] primtransf (mvpraymarch.py:355-360) and chlast=False give the same image
    packed = torch.cat([s["primpos"][:, :, None], s["primrot"], s["primscale"][:, :, None]], dim=2).contiguous()
    a = ops.mvpraymarch(rp, rd, s["stepsize"], tm, prim, s["template"], None)
    b = ops.mvpraymarch(rp, rd, s["stepsize"], tm, packed, s["template"], None)
    c = ops.mvpraymarch(rp, rd, s["stepsize"], tm, prim, s["template"].permute(0, 1, 5, 2, 3, 4).contiguous(), None,
                        chlast=False)
    assert torch.equal(a, b) and torch.equal(a, c)
