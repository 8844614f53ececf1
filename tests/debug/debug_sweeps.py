"""GPU debugging aid: lane-independent forward sweep vs the slot-synchronous one (debug build), full image and a
sub-rectangle; prints how many rays differ and where."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ava256_amd as ops
from ava256_amd import _lib
from ava256_amd.scene import make_scene

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DBG = os.path.join(ROOT, "build_variants", "libmvp_dbg.so")


def render(s, sl=slice(None)):
    rp, rd, tm = ops.compute_raydirs(s["campos"][sl], s["camrot"][sl], s["focal"][sl], s["princpt"][sl], s["pixelcoords"][sl], s["volradius"])
    with torch.no_grad():
        return ops.mvpraymarch(rp, rd, s["stepsize"], tm, (s["primpos"][sl], s["primrot"][sl], s["primscale"][sl]), s["template"][sl], None)


def report(name, a, b):
    d = (a - b).abs().amax(-1)
    n = int((d > 0).sum())
    print(name, "rays differing", n, "of", d.numel(), "max abs diff %.3e" % float(d.max()))
    if n:
        idx = torch.nonzero(d > 0)[:8]
        for i in idx.tolist():
            print("   ", i, a[tuple(i)].tolist(), b[tuple(i)].tolist())


for gain in (1.0, 20.0):
    s = make_scene(6, 512, 512, 4096, device="cuda", seed=1112, alpha_gain=gain)
    full_fast = render(s)
    y0, y1, x0, x1 = 101, 367, 59, 402
    sub = dict(s)
    sub["pixelcoords"] = s["pixelcoords"][:, y0:y1, x0:x1].contiguous()
    part_fast = render(sub, slice(3, 6))
    os.environ["MVP_DEBUG_SLOT_SWEEP"] = "1"
    _lib.use_library(DBG)
    full_slot = render(s)
    part_slot = render(sub, slice(3, 6))
    del os.environ["MVP_DEBUG_SLOT_SWEEP"]
    _lib.use_library(None)
    print("alpha gain", gain)
    report("  full: fast vs slot", full_fast, full_slot)
    report("  part: fast vs slot", part_fast, part_slot)
    report("  fast: part vs full", part_fast, full_fast[3:6, y0:y1, x0:x1])
    report("  slot: part vs full", part_slot, full_slot[3:6, y0:y1, x0:x1])
