"""tools/diag_xcd_balance.py -- how evenly the C2 bench scene loads the 8 XCDs under the whole-image mapping (XCD x renders images x, x + 8, ...):
list entries, hitting rays and accumulated alpha per image, summed per XCD.  Measured (round 6): max / mean over the XCDs 1.009 / 1.0005 / 1.0001 -- the
synthetic cameras see the same head, so a dynamic image queue has nothing to balance here (a real multi-camera batch may differ)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import __graft_entry__
import ava256_amd as ops
from ava256_amd import _hooks
from ava256_amd.scene import make_scene
N, H, W, K = 80, 512, 512, 4096
dev = torch.device("cuda:0")
s = make_scene(N, H, W, K, device=dev, seed=1112)
rp, rd, tm = ops.compute_raydirs(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"], s["volradius"])
t = {k: s[k].clone().requires_grad_(True) for k in ("primpos", "primrot", "primscale", "template")}
_hooks.keep_raysat = True
rgba = ops.mvpraymarch(rp, rd, s["stepsize"], tm, (t["primpos"], t["primrot"], t["primscale"]), t["template"], None)
torch.cuda.synchronize()
cnt = (_hooks.last_pl_count[: N * K] & 0x3fffffff).view(N, K).sum(1).float()   # list entries per image
hit = (rgba[..., 3] > 0).view(N, -1).float().sum(1)                              # hitting rays per image
alpha = rgba[..., 3].view(N, -1).sum(1)                                          # ~ samples weight
for name, v in (("entries", cnt), ("hit rays", hit), ("sum alpha", alpha)):
    per_xcd = v.view(10, 8).sum(0)
    print(name, "per image min/mean/max %.0f %.0f %.0f" % (v.min(), v.mean(), v.max()),
          "| per XCD max/mean %.4f" % (per_xcd.max() / per_xcd.mean()), [round(float(x / per_xcd.mean()), 3) for x in per_xcd])
