"""tools/bench_assemble_frames.py [library.so] -- the frame-broadcast hand-off kernels at C3 (F = 4, 16384 primitives) and C2
(F = 80, 4096 primitives): ms and GB/s forward / backward."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ava256_amd import _lib  # noqa: E402
if len(sys.argv) > 1:
    _lib.use_library(os.path.abspath(sys.argv[1]))
from ava256_amd.assemble import assemble_template_frames  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


out = {}
for name, F, nh in (("C3", 4, 128), ("C2", 80, 64)):
    B = 8
    S = nh * B
    tex = torch.randn(1, 3 * B, S, S, device="cuda", requires_grad=True)
    op = torch.randn(1, B, S, S, device="cuda", requires_grad=True)
    gain = (1 + 0.1 * torch.randn(F, device="cuda")).requires_grad_(True)
    gout = torch.randn(F, nh * nh, B, B, B, 4, device="cuda")
    t_f = timeit(lambda: assemble_template_frames(tex, op, gain, nh * nh, B))

    def fb():
        tex.grad = op.grad = gain.grad = None
        assemble_template_frames(tex, op, gain, nh * nh, B).backward(gout)
    t_fb = timeit(fb)
    slab = 4 * B * S * S * 4
    out[name] = {"fwd_ms": t_f, "fwd_GBps": (F + 1) * slab / t_f / 1e6, "bwd_ms": t_fb - t_f, "bwd_GBps": (F + 2) * slab / (t_fb - t_f) / 1e6}
    del tex, op, gain, gout
print(json.dumps(out))
