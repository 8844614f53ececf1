"""Time the primitive-placement kernel (barycentric half of row N2) against the eager form the reference uses: a full
1024 x 1024 position map from three index_selects, then strided reads (assembler.py:118-122,180-206), forward+backward.
Usage: python tools/bench_placement.py [B]"""
import sys, time, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import make_placement_inputs
from ava256_amd.placement import prim_placement

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
geo_np, idx_np, bar_np, volradius, _ = make_placement_inputs(16384, B=B)
geo = torch.from_numpy(geo_np).to(dev).requires_grad_(True)
idx, bar = torch.from_numpy(idx_np).to(dev), torch.from_numpy(bar_np).to(dev)
T = idx.shape[0]
i0, i1, i2 = [idx[:, :, c].reshape(-1) for c in range(3)]


def eager():
    post = (bar[:, :, 0, None] * geo.index_select(1, i0).reshape(-1, T, T, 3) +
            bar[:, :, 1, None] * geo.index_select(1, i1).reshape(-1, T, T, 3) +
            bar[:, :, 2, None] * geo.index_select(1, i2).reshape(-1, T, T, 3)).permute(0, 3, 1, 2) / volradius
    primpos = post[:, :, 4::8, 4::8].permute(0, 2, 3, 1).contiguous().view(B, 16384, 3)
    du = (post[:, :, :, 1:] - post[:, :, :, :-1])[:, :, 4::8, 4::8].permute(0, 2, 3, 1)
    dv = (post[:, :, 1:, :] - post[:, :, :-1, :])[:, :, 4::8, 4::8].permute(0, 2, 3, 1)
    return primpos, du, dv


def hip():
    return prim_placement(geo, idx, bar, volradius, 16384)


for name, fn in (("hip  ", hip), ("eager", eager)):
    for _ in range(3):
        geo.grad = None
        a, b, c = fn(); (a.sum() + b.sum() + c.sum()).backward()
    torch.cuda.synchronize()
    tf, tb = [], []
    for _ in range(10):
        geo.grad = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        a, b, c = fn(); torch.cuda.synchronize(); t1 = time.perf_counter()
        (a.sum() + b.sum() + c.sum()).backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
        tf.append(t1 - t0); tb.append(t2 - t1)
    print("%s B=%d  forward %7.3f ms   backward (incl. the three sums) %7.3f ms" % (name, B, min(tf) * 1e3, min(tb) * 1e3))
