"""Time the background MLP (forward, forward + backward) at B x H x W: fused MFMA kernels vs eager bf16 autocast.
usage: python tools/bench_bgmlp_fused.py [B H W [variant library]]   (with a build_variants/ library: fused only -- timing
experiments compute wrong results by construction)"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import __graft_entry__  # noqa: F401
from ava256_amd import _lib
VARIANT = sys.argv[4] if len(sys.argv) >= 5 else None
if VARIANT:
    _lib.use_library(os.path.abspath(VARIANT))
from ava256_amd.trainloop import BackgroundMLPStandIn

B, H, W = [int(x) for x in (sys.argv[1:4] if len(sys.argv) >= 4 else (4, 512, 512))]
gen = torch.Generator().manual_seed(0)
cam, idx = torch.randint(0, 5, (B,), generator=gen).cuda(), torch.randint(0, 3, (B,), generator=gen).cuda()
ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
sc = torch.stack([xs, ys], -1)[None].expand(B, H, W, 2).contiguous().cuda()
gout = torch.randn(B, 3, H, W, generator=gen).cuda()
flop_fwd = 2.0 * B * H * W * (120 * 256 + 4 * 256 * 256 + 256 * 3)


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n


out = dict(B=B, H=H, W=W, gflop_fwd=round(flop_fwd * 1e-9, 1))
if VARIANT:
    out["library"] = VARIANT
for name, fused in ((("fused", True),) if VARIANT else (("fused", True), ("eager", False))):
    m = BackgroundMLPStandIn(5, 3, fused=fused).cuda()

    def fwd():
        with torch.no_grad():
            return m(cam, idx, sc)

    def fb():
        for p in m.parameters():
            p.grad = None
        m(cam, idx, sc).backward(gout)

    tf, tfb = timeit(fwd), timeit(fb)
    out[name] = dict(fwd_ms=round(tf, 3), fwd_tflops=round(flop_fwd / tf * 1e-9, 1), fwd_bwd_ms=round(tfb, 3),
                     fwd_bwd_tflops=round(3 * flop_fwd / tfb * 1e-9, 1))
print(json.dumps(out))
