# Test infrastructure (uses the oracle as the checker): reproduces the heavy-scene traversal fallback outside pytest.
import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import ava256_amd as ops
from ava256_amd import _hooks
from ava256_amd.scene import make_scene
from oracle.mvp_oracle import Oracle
from helpers import scene_rays, to_dev, npf
o = Oracle("f64")
def run(K, scale_mul, again, H=24):
    s = make_scene(1, H, H, K, device="cpu", seed=99, alpha_gain=again, order=os.environ.get("ORDER","uvgrid"))
    s["primscale"] = s["primscale"] * scale_mul
    rp, rd, tm = scene_rays(o, s)
    a = (rp, rd, s["stepsize"], tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy(), s["template"].numpy())
    ref, sat, st = o.march_forward(*a)
    diag = torch.zeros(8, dtype=torch.int32, device="cuda"); _hooks.set_diag_buffer(diag)
    with torch.no_grad():
        out = ops.mvpraymarch(to_dev(rp), to_dev(rd), s["stepsize"], to_dev(tm), (s["primpos"].cuda(), s["primrot"].cuda(), s["primscale"].cuda()), s["template"].cuda(), None)
    torch.cuda.synchronize(); d = _hooks.read_diag(); _hooks.set_diag_buffer(None)
    err = np.abs(npf(out) - ref).max(-1)[0]
    print("K", K, "scale", scale_mul, "force_dfs", os.environ.get("MVP_DEBUG_FORCE_DFS"), d, "oracle list/ray", st["list_len_sum"] / max(1, st["rays_hit"]))
    print("  max err", err.max(), "n bad", (err > 1e-3 * max(1, np.abs(ref).max())).sum(), "of", err.size, " ref max", np.abs(ref).max())
    bad = np.argwhere(err > 1e-3 * max(1, np.abs(ref).max()))[:5]
    for (y, x) in bad: print("   bad px", y, x, "ref", ref[0, y, x], "got", npf(out)[0, y, x])
run(int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3]))
