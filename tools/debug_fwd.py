"""Forward-only smoke over a few (N, K) on the GPU; prints before/after each so a crash can be located."""
import sys, torch
sys.path.insert(0, ".")
import ava256_amd as ops
from ava256_amd.scene import make_scene
from ava256_amd.raydirs import compute_raydirs
for (N, K) in [(1, 8), (2, 8), (1, 512), (2, 512)]:
    s = make_scene(N, 64, 64, K, device="cuda", seed=7 + K, alpha_gain=1.0, slab=8)
    print("start", N, K, flush=True)
    rp, rd, tm = compute_raydirs(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"], s["volradius"])
    out = ops.mvpraymarch(rp, rd, float(s["stepsize"]), tm, (s["primpos"], s["primrot"], s["primscale"]), s["template"], None)
    torch.cuda.synchronize()
    print("ok", N, K, float(out.abs().sum()), flush=True)
