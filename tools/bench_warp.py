"""Time the algo-1 (warp-field) forward + backward with the primitive-centric and the ray-centric backward.
usage: python tools/bench_warp.py [N H W K]"""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import __graft_entry__  # noqa: F401
import ava256_amd as ops
from ava256_amd import _hooks
from ava256_amd.scene import make_scene
if os.environ.get("MVP_VARIANT_LIB"):   # timing A/B against a build_variants/ library (never the product path)
    from ava256_amd import _lib
    _lib.use_library(os.path.abspath(os.environ["MVP_VARIANT_LIB"]))

N, H, W, K = [int(x) for x in (sys.argv[1:5] if len(sys.argv) >= 5 else (4, 512, 512, 4096))]
s = make_scene(N, H, W, K, device="cuda", seed=3, alpha_gain=1.0)
g = torch.Generator(device="cuda").manual_seed(1)
lin = torch.linspace(-1, 1, 8, device="cuda")
zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
warp = (torch.stack([xx, yy, zz], -1)[None, None] + 0.05 * torch.randn(N, K, 8, 8, 8, 3, device="cuda", generator=g)).contiguous()
rp, rd, tm = ops.compute_raydirs(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"], s["volradius"])
out = {}
for mode in ("plain", "prim", "ray"):   # "plain": the same scene through algo 0 (no warp field) -- what the warp field costs
    handoff = _hooks.patched_handoff(ray_centric=(mode == "ray"))
    t = {k: s[k].clone().requires_grad_(True) for k in ("primpos", "primrot", "primscale", "template")}
    w = warp.clone().requires_grad_(True)
    gout = torch.randn(N, H, W, 4, device="cuda", generator=g)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    fw = bw = 0.0
    for it in range(4):
        for v in list(t.values()) + [w]:
            v.grad = None
        ev[0].record()
        with handoff:
            if mode == "plain":
                rgba = ops.mvpraymarch(rp, rd, s["stepsize"], tm, (t["primpos"], t["primrot"], t["primscale"]), t["template"], None)
            else:
                rgba = ops.mvpraymarch(rp, rd, s["stepsize"], tm, (t["primpos"], t["primrot"], t["primscale"]), t["template"], w, algo=1)
        ev[1].record()
        rgba.backward(gout)
        ev[2].record()
        torch.cuda.synchronize()
        if it > 0:
            fw += ev[0].elapsed_time(ev[1]) / 3
            bw += ev[1].elapsed_time(ev[2]) / 3
    out[mode] = dict(fwd_ms=round(fw, 3), bwd_ms=round(bw, 3), gw_norm=float(w.grad.norm()) if w.grad is not None else None,
                     gt_norm=float(t["template"].grad.norm()))
print(json.dumps(dict(N=N, H=H, W=W, K=K, **out)))
