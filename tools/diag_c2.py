"""Print the march diagnostics (list sizes etc.) for a bench workload. Usage: python tools/diag_c2.py [C2|C3|C1] [ncams]"""
import sys, torch
sys.path.insert(0, '/root/repo')
import ava256_amd as ops
from ava256_amd import _hooks
from ava256_amd.scene import make_scene
sys.path.insert(0, '/root/repo'); import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
N, H, W, K, slab = bench.WORKLOADS[wl]
if len(sys.argv) > 2: N = int(sys.argv[2])
s = make_scene(N, H, W, K, device="cuda", seed=1112, slab=slab)
diag = torch.zeros(8, dtype=torch.int32, device="cuda"); _hooks.set_diag_buffer(diag)
rp, rd, tm = ops.compute_raydirs(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"], s["volradius"])
for k in ("primpos", "primrot", "primscale", "template"): s[k].requires_grad_(True)
rgba = ops.mvpraymarch(rp, rd, s["stepsize"], tm, (s["primpos"], s["primrot"], s["primscale"]), s["template"], None)
torch.cuda.synchronize()
d = _hooks.read_diag(); _hooks.set_diag_buffer(None)
pk = N * ((H + 7) // 8) * ((W + 7) // 8)
print(wl, "N", N, d)
print("packets", pk, "hit frac", d["packets_hit"] / pk, "avg list", d["list_entries"] / max(1, d["packets_hit"]),
      "avg cand", d["candidates"] / max(1, d["packets_hit"]), "entries per prim", d["list_entries"] / (N * K))
print("alpha>0 frac", (rgba[..., 3] > 0).float().mean().item())
