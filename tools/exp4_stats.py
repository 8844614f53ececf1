"""With the MVP_EXP=4 library (tools/exp_variants.sh 4; bind it with ava256_amd._lib.use_library): LDS scatter statistics of the primitive-centric backward at a bench workload."""
import sys, torch
sys.path.insert(0, "."); import bench
import ava256_amd as ops
from ava256_amd import _hooks
from ava256_amd.scene import make_scene
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
N, H, W, K, slab = bench.WORKLOADS[wl]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
s = make_scene(N, H, W, K, device="cuda", seed=1112, slab=slab)
rp, rd, tm = ops.compute_raydirs(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"], s["volradius"])
for k in ("primpos", "primrot", "primscale", "template"): s[k].requires_grad_(True)
diag = torch.zeros(16, dtype=torch.int32, device="cuda"); _hooks.set_diag_buffer(diag)
rgba = ops.mvpraymarch(rp, rd, s["stepsize"], tm, (s["primpos"], s["primrot"], s["primscale"]), s["template"], None)
torch.cuda.synchronize(); diag.zero_()
rgba.backward(torch.randn_like(rgba)); torch.cuda.synchronize()
d = diag.cpu().tolist()
g, a, l = d[0], d[1], d[2]
print("groups", g, "lanes/group", l / g, "same-address max per group", a / g)
for h, nm in enumerate(["8,4 (current)", "8,5", "8,12", "8,20", "9,17", "9,5", "10,20", "12,3"]):
    print("  (y stride, z stride mod 32) =", nm, ": same-bank max per group", d[8 + h] / g)
