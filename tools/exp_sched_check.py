#!/usr/bin/env python3
"""tools/exp_sched_check.py <variant.so> -- the CU-local packet scheduler experiment (round 5; sources:
profiles/r05_fwd_cu_scheduler_experiment.patch): the forward through the variant library with
MVP_SCHED=1 must give the product's image, raysat and per-primitive packet counts bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

if __name__ == "__main__":
    from ava256_amd import _hooks, _lib
    import ava256_amd as ops
    from ava256_amd.scene import make_scene
    s = make_scene(10, 512, 512, 4096, device="cuda", seed=1112)

    def fwd():
        _hooks.keep_raysat = True
        rp, rd, tm = ops.compute_raydirs(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"], s["volradius"])
        t = {k: s[k].clone().requires_grad_(True) for k in ("primpos", "primrot", "primscale", "template")}
        rgba = ops.mvpraymarch(rp, rd, s["stepsize"], tm, (t["primpos"], t["primrot"], t["primscale"]), t["template"], None)
        sat, cnt = _hooks.last_raysat, _hooks.last_pl_count
        _hooks.keep_raysat = False
        g = torch.ones_like(rgba)
        rgba.backward(g)
        torch.cuda.synchronize()
        return rgba.detach(), sat, cnt[: 10 * 4096].clone(), t["template"].grad

    a = fwd()
    _lib.use_library(os.path.abspath(sys.argv[1]))
    os.environ["MVP_SCHED"] = "1"
    b = fwd()
    names = ("rgba", "raysat", "packet counts", "grad_template")
    ok = [bool(torch.equal(x, y)) for x, y in zip(a, b)]
    print(dict(zip(names, ok)))
    raise SystemExit(0 if all(ok) else 1)
