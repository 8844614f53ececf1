"""tools/diag_two_pass_marks.py [workload] -- how many primitives of a TRAINING iteration's backward go to the two-pass (residual)
kernel or to the ray-centric kernel, and what the upstream gradient of the march looks like (why)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ava256_amd import _hooks  # noqa: E402
from ava256_amd.trainloop import (BackgroundMLPStandIn, CodeEncoderStandIn, ColorCalStandIn, RaymarchTrainModel,  # noqa: E402
                                  SlabDecoderStandIn, Trainer, make_training_batch)

workload = sys.argv[1] if len(sys.argv) > 1 else "C2"
with_bg = len(sys.argv) > 2 and sys.argv[2] == "bg"     # (round 6) ... with the background MLP in the decode tail
dev = torch.device("cuda:0")
N, H, W, K, slab = bench.WORKLOADS[workload]
batch, volradius = make_training_batch(N, H, W, K, dev, seed=1112, ncams=80, nident=4, target_decoder=SlabDecoderStandIn(K, slab, seed=9))
model = RaymarchTrainModel(SlabDecoderStandIn(K, slab, seed=1), volradius, colorcal=ColorCalStandIn(80, 4),
                           bgmodel=BackgroundMLPStandIn(80, 4) if with_bg else None, encoder=CodeEncoderStandIn()).to(dev)
tr = Trainer(model)
_hooks.keep_raysat = True
grads = {}
import importlib  # noqa: E402
mm = importlib.import_module('ava256_amd.mvpraymarch')
orig = mm._backward_impl


def spy(ctx, g):
    grads["g"] = g.detach().clone()
    return orig(ctx, g)


mm._backward_impl = spy
for it in range(4):
    tr.step(batch)
torch.cuda.synchronize()
c = _hooks.last_pl_count[: N * K].cpu()
cnt = (c & 0x3fffffff)
print("primitives", N * K, "listed", int((cnt > 0).sum()), "precise", int(((c >> 30) & 1).sum()), "dead", int(((c >> 31) & 1).sum()))
g = grads["g"]                      # [N,H,W,4]
m = g.abs().amax(-1)                # per ray
P = m.reshape(N, H // 8, 8, W // 8, 8).permute(0, 1, 3, 2, 4).reshape(N, H // 8, W // 8, 64)
pmax, pmin_nz = P.amax(-1), torch.where(P > 0, P, torch.full_like(P, float("inf"))).amin(-1)
hit = pmax > 0
ratio = (pmax / pmin_nz)[hit]
print("rays with g == 0: %.3f;  per-packet max/min(nonzero) ratio: median %.1f  p90 %.1f  p99 %.1f  max %.3g; packets with ratio > 256: %.4f"
      % (float((m == 0).float().mean()), float(ratio.median()), float(ratio.quantile(0.9)), float(ratio.quantile(0.99)), float(ratio.max()),
         float((ratio > 256).float().mean())))
print("global max |g| %.3g, median nonzero %.3g; per-channel max %s" % (float(m.max()), float(m[m > 0].median()), g.abs().amax((0, 1, 2)).tolist()))
# (round 6) the marked primitives themselves: list lengths (ray packets per primitive) and where they sit in the dispatch order
marked = ((c >> 30) & 1).bool()
if bool(marked.any()):
    lens = cnt[marked].float()
    ks = torch.nonzero(marked).flatten() % K
    print("two-pass primitives: list length min %d median %d max %d; all primitives: median %d p99 %d max %d; distinct k among the marked: %d"
          % (int(lens.min()), int(lens.median()), int(lens.max()), int(cnt.float().median()), int(cnt.float().quantile(0.99)),
             int(cnt.max()), int(ks.unique().numel())))
    img = torch.nonzero(marked).flatten() // K
    print("marked per image: min %d max %d" % (int(torch.bincount(img, minlength=N).min()), int(torch.bincount(img, minlength=N).max())))
