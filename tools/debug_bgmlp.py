import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import __graft_entry__  # noqa
from ava256_amd.trainloop import BackgroundMLPStandIn
B, H, W = 2, 96, 80
gen = torch.Generator().manual_seed(5)
cam, idx = torch.randint(0, 5, (B,), generator=gen).cuda(), torch.randint(0, 3, (B,), generator=gen).cuda()
sc = (torch.rand(B, H, W, 2, generator=gen) * 2 - 1).cuda()
gout = torch.randn(B, 3, H, W, generator=gen).cuda()
res = {}
for name, fused, dt in (("fp32", False, None), ("eager_bf16", False, torch.bfloat16), ("fused", True, torch.bfloat16)):
    m = BackgroundMLPStandIn(5, 3, autocast_dtype=dt, fused=fused).cuda()
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=torch.Generator().manual_seed(p.numel())).cuda())
    bg = m(cam, idx, sc)
    (bg * gout).sum().backward()
    res[name] = (bg.detach().double().cpu().numpy(), {k: p.grad.detach().double().cpu().numpy() for k, p in m.named_parameters()})
ref = res["fp32"]
for name in ("eager_bf16", "fused"):
    bg, g = res[name]
    print(name, "out max err / spread: %.2e" % (np.abs(bg - ref[0]).max() / np.abs(ref[0] - 100).max()))
    for k in g:
        a, b = g[k].ravel(), ref[1][k].ravel()
        print("   %-18s cos %.5f  rel %.3e" % (k, a @ b / np.linalg.norm(a) / np.linalg.norm(b), np.linalg.norm(a - b) / np.linalg.norm(b)))
# timing split of the fused backward at 4 x 512 x 512
from ava256_amd import bgmlp as bm, _lib
from ava256_amd._tensors import ptr, stream_ptr
B, H, W = 4, 512, 512
P = B * H * W
acts = torch.randn(5, P, 256, device="cuda").to(torch.bfloat16)
dz = torch.empty_like(acts)
whT = torch.randn(4, 256, 256, device="cuda").to(torch.bfloat16)
w6 = torch.randn(3, 256, device="cuda")
go = torch.randn(B, 3, H, W, device="cuda")
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n
print("bwd kernel   %.3f ms" % t(lambda: _lib.check(_lib.get_lib().mvp_bgmlp_backward(B, H * W, ptr(go), ptr(acts), ptr(whT), ptr(w6), ptr(dz), stream_ptr(go.device)), "b")))
print("wgrad 256x256 (chunked bmm) %.3f ms each" % t(lambda: bm._wgrad(dz[1], acts[0])))
print("bias colsum  %.3f ms each" % t(lambda: dz[1].sum(0, dtype=torch.float32)))
print("bias1 sum    %.3f ms" % t(lambda: dz[0].view(B, H * W, 256).sum(1, dtype=torch.float32)))
sc4 = torch.rand(P, 2, device="cuda")
print("posenc       %.3f ms" % t(lambda: bm.positional_encoding(sc4).to(torch.bfloat16)))
x0 = bm.positional_encoding(sc4).to(torch.bfloat16)
print("wgrad 256x40 %.3f ms" % t(lambda: bm._wgrad(dz[0], x0)))
g6 = torch.randn(P, 3, device="cuda").to(torch.bfloat16)
print("wgrad 3x256  %.3f ms" % t(lambda: bm._wgrad(g6, acts[4])))
print("whT transpose+cast %.3f ms" % t(lambda: whT.transpose(1, 2).contiguous()))
