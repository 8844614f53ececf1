#!/usr/bin/env python3
"""tools/diag_train_scene.py [workload] -- the forward's diag counters (hit packets, packets on the slot-synchronous sweep, list
entries, BVH candidates) and hand-off statistics (list length per primitive) of (a) the bench scene and (b) the scene a
TRAINING iteration renders at the same shape -- why train.C2's march kernels take 7-12 % longer than the bench's."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import __graft_entry__  # noqa: F401
import ava256_amd as ops
from ava256_amd import _hooks
from ava256_amd.scene import make_scene
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
N, H, W, K, slab = bench.WORKLOADS[wl]
dev = torch.device("cuda:0")
out = {}


def counters(run):
    diag = torch.zeros(8, dtype=torch.int32, device=dev)
    _hooks.set_diag_buffer(diag)
    _hooks.keep_raysat = True
    run()
    torch.cuda.synchronize()
    d = _hooks.read_diag()
    cnt = _hooks.last_pl_count
    sat = _hooks.last_raysat
    _hooks.keep_raysat = False
    _hooks.set_diag_buffer(None)
    c = (cnt[: N * K] & 0x3fffffff).float()
    d.update(list_len_mean=float(c.mean()), list_len_p50=float(c.median()), list_len_p99=float(c.quantile(0.99)),
             list_len_max=float(c.max()), prims_with_more_than_10=float((c > 10).float().mean()),
             saturated_ray_fraction=float((sat[..., 0] > -1).float().mean()) if sat is not None else None)
    d["slowpath_fraction"] = d["slowpath_packets"] / max(1, d["packets_hit"])
    d["entries_per_hit_packet"] = d["list_entries"] / max(1, d["packets_hit"])
    d["candidates_per_hit_packet"] = d["candidates"] / max(1, d["packets_hit"])
    _hooks.last_raysat = _hooks.last_pl_count = None
    return d


# (a) the bench scene
s = make_scene(N, H, W, K, device=dev, seed=1112, slab=slab)
rp, rd, tm = ops.compute_raydirs(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"], s["volradius"])
t = {k: s[k].clone().requires_grad_(True) for k in ("primpos", "primrot", "primscale", "template")}
out["bench_scene"] = counters(lambda: ops.mvpraymarch(rp, rd, s["stepsize"], tm, (t["primpos"], t["primrot"], t["primscale"]), t["template"], None))
del s, rp, rd, tm, t
torch.cuda.empty_cache()

# (b) the training scene: the iteration of bench.train_leg
from ava256_amd.trainloop import (CodeEncoderStandIn, ColorCalStandIn, RaymarchTrainModel, SlabDecoderStandIn, Trainer,
                                  make_training_batch)
ncams, nident = 80, 4
batch, volradius = make_training_batch(N, H, W, K, dev, seed=1112, ncams=ncams, nident=nident,
                                       target_decoder=SlabDecoderStandIn(K, slab, seed=9))
model = RaymarchTrainModel(SlabDecoderStandIn(K, slab, seed=1), volradius, colorcal=ColorCalStandIn(ncams, nident),
                           bgmodel=None, encoder=CodeEncoderStandIn()).to(dev)
tr = Trainer(model, ddp=False)
for _ in range(3):
    tr.step(batch)
out["train_scene_iteration_3"] = counters(lambda: tr.step(batch))
print(json.dumps(out, indent=1))
