"""tools/diag_graph_capture.py -- which operator of the training iteration refuses hipGraph capture (forward + backward of each
piece captured on its own; prints ok / the error)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ava256_amd.trainloop import (CodeEncoderStandIn, ColorCalStandIn, RaymarchTrainModel, SlabDecoderStandIn,  # noqa: E402
                                  make_training_batch)

dev = "cuda"
K = 256
batch, volradius = make_training_batch(2, 64, 64, K, dev, seed=5, target_decoder=SlabDecoderStandIn(K, seed=9))
dec = SlabDecoderStandIn(K, seed=1).to(dev)
cc = ColorCalStandIn(80, 4).to(dev)
enc = CodeEncoderStandIn().to(dev)
model = RaymarchTrainModel(SlabDecoderStandIn(K, seed=1), volradius, colorcal=ColorCalStandIn(80, 4), encoder=CodeEncoderStandIn()).to(dev)


def attempt(name, fn, params):
    for _ in range(2):
        for p in params:
            p.grad = None
        fn().backward()
    torch.cuda.synchronize()
    for p in params:
        p.grad = None
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            fn().backward()
        g.replay()
        torch.cuda.synchronize()
        print("%-28s ok" % name, flush=True)
    except Exception as e:
        print("%-28s FAILED: %s" % (name, str(e).splitlines()[0]), flush=True)
        torch.cuda.synchronize()


sched = {"running_avg_scale": False, "use_gt_geo": False, "residuals_weight": 1.0}
code = batch["code"]


def dec_out(keys):
    def f():
        o = dec(code, schedule=sched)
        return sum(o[k].sum() for k in keys)
    return f


attempt("decoder: template", dec_out(["template"]), list(dec.parameters()))
attempt("decoder: pose", dec_out(["primpos", "primrot", "primscale"]), list(dec.parameters()))
attempt("decoder: verts", dec_out(["verts"]), list(dec.parameters()))
attempt("colorcal", lambda: cc(batch["image"], batch["camindex"], batch["idindex"]).sum(), list(cc.parameters()))
attempt("encoder", lambda: sum(v.sum() for v in enc(code, batch["noise"]).values() if torch.is_tensor(v)) if isinstance(enc(code, batch["noise"]), dict)
        else sum(t.sum() for t in enc(code, batch["noise"]) if torch.is_tensor(t)), list(enc.parameters()))


def full(target):
    def f():
        kw = {"target": batch["image"]} if target else {}
        o = model(batch["camrot"], batch["campos"], batch["focal"], batch["princpt"], batch["pixelcoords"], code, schedule=sched,
                  camindex=batch["camindex"], idindex=batch["idindex"], gt_verts=batch["verts"], noise=batch["noise"], **kw)
        return o["irgbl1_sum"] if target else o["irgbrec"].sum()
    return f


attempt("model (no fused L1)", full(False), list(model.parameters()))
attempt("model (fused tail + L1)", full(True), list(model.parameters()))
