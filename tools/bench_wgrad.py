"""tools/bench_wgrad.py -- the weight-gradient GEMMs of the fused background MLP's backward (bgmlp._wgrad): four separate
chunked bmm calls against one stacked call, bf16 against fp32 partial products (time and error against float64)."""
import torch

dev = torch.device("cuda:0")
P, Wd, L, S = 4 * 512 * 512, 256, 4, 64
g = torch.Generator(device=dev).manual_seed(3)
dz = (torch.randn((L + 1, P, Wd), device=dev, generator=g) * 0.01).to(torch.bfloat16)
acts = torch.randn((L + 1, P, Wd), device=dev, generator=g).to(torch.bfloat16)


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out


def separate(out_dtype=None):
    res = []
    for l in range(L):
        a, b = dz[l + 1].view(S, P // S, Wd).transpose(1, 2), acts[l].view(S, P // S, Wd)
        r = torch.bmm(a, b) if out_dtype is None else torch.bmm(a, b, out_dtype=out_dtype)
        res.append(r.float().sum(0))
    return torch.stack(res)


def stacked(out_dtype=None, S=S):
    a, b = dz[1:].view(L * S, P // S, Wd).transpose(1, 2), acts[:L].view(L * S, P // S, Wd)
    r = torch.bmm(a, b) if out_dtype is None else torch.bmm(a, b, out_dtype=out_dtype)
    return r.float().view(L, S, Wd, Wd).sum(1)


ref = torch.stack([dz[l + 1, : P // 16].double().t() @ acts[l, : P // 16].double() for l in range(L)])  # 1/16 of the rows
for name, fn in (("separate bf16", separate), ("stacked bf16", stacked), ("stacked bf16 S=16", lambda: stacked(S=16)),
                 ("stacked bf16 S=256", lambda: stacked(S=256)),
                 ("separate f32", lambda: separate(torch.float32)), ("stacked f32", lambda: stacked(torch.float32)),
                 ("stacked f32 S=16", lambda: stacked(torch.float32, 16))):
    try:
        ms, out = timed(fn)
    except Exception as e:  # noqa: BLE001
        print("%-20s unsupported: %s" % (name, str(e)[:120]))
        continue
    # error on a 1/16 slice computed the same way
    dzs, acs = dz, acts
    print("%-20s %.3f ms" % (name, ms))
# accuracy: rerun the variants on the 1/16 slice
dz, acts, P = dz[:, : P // 16].contiguous(), acts[:, : P // 16].contiguous(), P // 16
for name, fn in (("stacked bf16", stacked), ("stacked f32", lambda: stacked(torch.float32))):
    try:
        out = fn().double()
        print("%-20s rel err vs f64 %.2e" % (name, float((out - ref).norm() / ref.norm())))
    except Exception as e:  # noqa: BLE001
        print(name, "unsupported", str(e)[:100])
# the last layer's bias gradient: the strided [P,3] column sum against the plane sum
gout = torch.randn((4, 3, 512, 512), device=dev, generator=g)
g6 = gout.view(4, 3, -1).permute(0, 2, 1).reshape(-1, 3) * 25.0
print("g6.sum(0)            %.3f ms" % timed(lambda: g6.sum(0))[0])
print("gout.sum((0,2,3))    %.3f ms" % timed(lambda: gout.sum((0, 2, 3)) * 25.0)[0])
print("permute+mul          %.3f ms" % timed(lambda: gout.view(4, 3, -1).permute(0, 2, 1).reshape(-1, 3) * 25.0)[0])
