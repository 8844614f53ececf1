#!/usr/bin/env python3
"""tools/measure_reference_cpu.py -- time the REFERENCE's own CPU-runnable statement of the raymarch in this build container
and record it, stamped, in profiles/reference_cpu.json (bench.py carries it as `reference_cpu`).

What is timed: the dense pure-PyTorch raymarch + autograd backward of /root/reference/extensions/mvpraymarch/mvpraymarch.py
(`gradcheck`, lines 553-641) on its STOCK scene (N=2, 65x65 rays, K=64 primitives, 32^3 slabs, fp32), executed from the
mounted file with "cuda" redirected to "cpu" -- the reference prints its own forward / backward seconds ("pytime"), which is
what is parsed.  The reference's Python cannot travel to the GPU box, so this is a build-container number (core count
recorded), carried beside the live `cpu_baseline` that bench.py measures on the GPU box's host.
(round 6) The reference `Autoencoder` forward / backward on CPU is RE-MEASURED as well (it was carried over from the survey until
round 5): the reference's own modules, imported from the mounted tree and assembled as its `utils.get_autoencoder` assembles them
(utils.py:77-113: K = 128^2 primitives of 8^3, 1024^2 UV maps, 7306 vertices), with recording stand-ins for its two native
modules -- the raymarch itself is stubbed, as in the survey -- and random arrays of the right shape for the UV index maps and the
mesh topology (`igl` / `trimesh`, which the real `create_uv_baridx` needs, are not installed).  Batch 1, 128 x 128 render.
Runs only where /root/reference is mounted.  ~2 minutes."""
import contextlib
import io
import json
import os
import re
import subprocess
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def dense_oracle_seconds():
    sys.modules["mvpraymarchlib"] = types.ModuleType("mvpraymarchlib")  # CUDA module stand-in
    src = open(os.path.join(REF, "extensions/mvpraymarch/mvpraymarch.py")).read()
    for a, b in (('"cuda"', '"cpu"'), ("torch.cuda.synchronize()", "pass"), ("from . import mvpraymarchlib", "import mvpraymarchlib")):
        assert a in src, a
        src = src.replace(a, b)
    ns = {"__name__": "refmvp"}
    exec(compile(src, "refmvp", "exec"), ns)

    class Done(Exception):
        pass

    def stop(*a, **k):   # the CUDA entry point: the dense statement has run by the time gradcheck calls it
        raise Done()

    ns["mvpraymarch"] = stop
    buf = io.StringIO()
    t0 = time.time()
    with contextlib.redirect_stdout(buf):
        try:
            ns["gradcheck"](usebvh="fixedorder", sortprims=False, maxhitboxes=512, synchitboxes=True, dowarp=False,
                            chlast=True, fadescale=8.0, fadeexp=8.0, accum=0, algo=0, griddim=3)
        except Done:
            pass
    total = time.time() - t0
    m = re.search(r"pytime\s+([0-9.eE+-]+)\s+([0-9.eE+-]+)\s+([0-9.eE+-]+)", buf.getvalue())
    assert m, buf.getvalue()[-500:]
    return float(m.group(1)), float(m.group(2)), total


def autoencoder_seconds(reps=3, H=128, W=128):
    """Forward and backward seconds of the reference Autoencoder on this container's CPU cores (median of `reps` timed
    iterations after one untimed iteration with running_avg_scale=True), its parameter count and the
    native calls one forward + backward makes."""
    import numpy as np
    sys.path.insert(0, REF)
    calls = []
    mv = types.ModuleType("mvpraymarchlib")          # mvpraymarch.cpp:398-405, positional signatures of SURVEY.md 8(b)

    def raymarch_forward(raypos, raydir, stepsize, tminmax, sortedobjid, nodechildren, nodeaabb, primpos, primrot, primscale,
                         template, warp, rayrgba, raysat, rayterm, *rest):
        calls.append("raymarch_forward")
        rayrgba.zero_()
        if raysat is not None:
            raysat.fill_(-1.0)

    mv.compute_aabb = lambda *a: calls.append("compute_aabb")
    mv.raymarch_forward = raymarch_forward
    mv.raymarch_backward = lambda *a: calls.append("raymarch_backward")     # (gradients stay the zeros the glue pre-filled)
    mv.compute_morton = mv.build_tree = lambda *a: None
    ut = types.ModuleType("utilslib")                # utils.cpp:134-137

    def compute_raydirs_forward(viewpos, viewrot, focal, princpt, pixelcoords, W_, H_, volradius, raypos, raydir, tminmax):
        calls.append("compute_raydirs_forward")
        raypos.zero_(), raydir.zero_(), tminmax.zero_()

    ut.compute_raydirs_forward = compute_raydirs_forward
    ut.compute_raydirs_backward = lambda *a: None
    sys.modules["mvpraymarchlib"], sys.modules["utilslib"] = mv, ut
    import models.autoencoder as aemodel
    import models.bg.mlp2d as bglib
    import models.bottlenecks.vae as vae
    import models.colorcals.colorcal as colorcalib
    import models.decoders.assembler as decoderlib
    import models.encoders.expression as expression_encoder_lib
    import models.encoders.identity as identity_encoder_lib
    import models.raymarchers.mvpraymarcher as raymarcherlib
    assert aemodel.__file__.startswith(REF), aemodel.__file__      # the reference's modules, not this repository's shims

    rng = np.random.default_rng(0)
    nv, res = 7306, 1024
    uv_idx = rng.integers(0, nv, size=(3, res, res)).astype(np.int64)
    uv_bary = rng.random((3, res, res)).astype(np.float32)
    vt = rng.random((nv, 2)).astype(np.float32)
    vi = rng.integers(0, nv, size=(14000, 3)).astype(np.int32)
    vti = vi.copy()
    torch.manual_seed(0)
    volradius = 256.0
    decoder = decoderlib.DecoderAssembler(vt=vt, vi=vi, vti=vti, idxim=uv_idx, barim=uv_bary, vertmean=torch.zeros(nv, 3),
                                          vertstd=1.0, volradius=volradius, nprims=128 * 128, primsize=(8, 8, 8))
    ae = aemodel.Autoencoder(identity_encoder=identity_encoder_lib.IdentityEncoder(uv_idx, uv_bary, wsize=128),
                             expression_encoder=expression_encoder_lib.ExpressionEncoder(uv_idx, uv_bary),
                             bottleneck=vae.VAE_bottleneck(64, 16), decoder_assembler=decoder,
                             raymarcher=raymarcherlib.Raymarcher(volradius), colorcal=colorcalib.Colorcal(80, 1),
                             bgmodel=bglib.BackgroundModelSimple(80, 1))
    nparams = sum(p.numel() for p in ae.parameters() if p.requires_grad)
    B = 1
    px, py = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    batch = dict(camrot=torch.eye(3)[None], campos=torch.tensor([[0.0, 0.0, -1430.0]]), focal=torch.full((B, 2), 5.0 * W),
                 princpt=torch.tensor([[W / 2.0, H / 2.0]]), modelmatrix=torch.eye(4)[None],
                 avgtex=torch.randn(B, 3, res, res), verts=torch.randn(B, nv, 3), neut_avgtex=torch.randn(B, 3, res, res),
                 neut_verts=torch.randn(B, nv, 3), target_neut_avgtex=torch.randn(B, 3, res, res),
                 target_neut_verts=torch.randn(B, nv, 3), pixelcoords=torch.from_numpy(np.stack((px, py), -1))[None],
                 idindex=torch.zeros(B, dtype=torch.long), camindex=torch.zeros(B, dtype=torch.long))
    fw, bw = [], []
    for it in range(reps + 1):
        for p_ in ae.parameters():
            p_.grad = None
        del calls[:]
        t0 = time.time()
        out = ae(**batch, running_avg_scale=(it == 0), output_set={"irgbrec"})
        t1 = time.time()
        out["irgbrec"].abs().mean().backward()
        t2 = time.time()
        if it > 0:
            fw.append(t1 - t0), bw.append(t2 - t1)
    fw.sort(), bw.sort()
    return fw[len(fw) // 2], bw[len(bw) // 2], nparams, list(calls)


if __name__ == "__main__":
    ae_fwd, ae_bwd, ae_params, ae_calls = autoencoder_seconds()
    fwd, bwd, total = dense_oracle_seconds()
    rays = 2 * 65 * 65
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    out = {
        "_measured_at_commit": commit, "_measured_on": time.strftime("%Y-%m-%d"), "_script": "tools/measure_reference_cpu.py",
        "where": "build container, %d CPU cores, torch %s on CPU; NOT the GPU box's host" % (os.cpu_count(), torch.__version__),
        "dense_raymarch_oracle": {"scene": "mvpraymarch.py gradcheck stock scene: N=2, 65x65, K=64, 32^3 slabs, fp32",
                                  "source": "reference's own 'pytime' line (mvpraymarch.py:636-637), executed from the mounted file",
                                  "fwd_s": fwd, "bwd_s": bwd, "fwd_rays_per_s": rays / fwd, "fwd_bwd_rays_per_s": rays / (fwd + bwd),
                                  "wall_s_of_the_run": total},
        "autoencoder": {"model": "reference Autoencoder (utils.py:77-113 assembly: identity + expression encoders, VAE bottleneck, "
                                 "DecoderAssembler with K = 16384 primitives of 8^3, Colorcal, BackgroundModelSimple), %.1f M "
                                 "trainable parameters, batch 1, 128x128 render, fp32, raymarch native calls stubbed" % (ae_params * 1e-6),
                        "fwd_s": ae_fwd, "bwd_s": ae_bwd, "threads": torch.get_num_threads(), "native_calls_per_iteration": ae_calls,
                        "source": "re-measured by this script (median of 3 iterations after one with running_avg_scale=True); "
                                  "the survey's numbers on the same 8 cores were 4.6 s / 1.5 s (BASELINE.md section 2)"},
    }
    path = os.path.join(ROOT, "profiles", "reference_cpu.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["dense_raymarch_oracle"]))
