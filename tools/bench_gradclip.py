"""Time the multi-tensor gradient hygiene passes (row N4) against the reference's eager statements, at ava-256's
parameter count (46.87 M floats = 187.5 MB) spread over ~600 tensors.  Usage: python tools/bench_gradclip.py"""
import sys, time, torch
sys.path.insert(0, ".")
from ava256_amd.gradclip import GradClipper

dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(3)
total, n = 46_870_000, 600
raw = torch.rand(n, generator=g) ** 3
sizes = (raw / raw.sum() * total).long().clamp(min=16).tolist()
params = [torch.nn.Parameter(torch.zeros(s, device=dev)) for s in sizes]
numel = sum(sizes)


for p in params:
    p.grad = torch.zeros_like(p)


def fill(nan_every=0):  # in place: the gradient tensors persist across iterations, as under zero_grad(set_to_none=False)
    for i, p in enumerate(params):
        p.grad.normal_(0.0, 0.01)
        if nan_every and i % nan_every == 0:
            p.grad[::97] = float("nan")


def eager(clip):
    for p in params:  # ddp-train.py:436-439
        p.grad.data[torch.isnan(p.grad.data)] = 0
        p.grad.data[torch.isinf(p.grad.data)] = 0
    return torch.nn.utils.clip_grad_norm_(params, clip)


clipper = GradClipper(dev)
for name, fn in (("hip   ", lambda c: clipper(params, c)), ("eager ", eager)):
    for clip, label in ((1.0, "clipping"), (1e9, "no clipping")):
        fill(nan_every=5); fn(clip); torch.cuda.synchronize()
        ts, es = [], []
        for _ in range(5):
            fill(nan_every=5); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter(); e0.record(); fn(clip); e1.record(); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0); es.append(e0.elapsed_time(e1) * 1e-3)
        t, e = min(ts), min(es)
        bytes_moved = numel * 4 * (3 if clip == 1.0 else 1)
        print("%s %-12s wall %8.3f ms  device %8.3f ms  %7.1f GB/s of the algorithmic %d bytes/elt" %
              (name, label, t * 1e3, e * 1e3, bytes_moved / e / 1e9, bytes_moved // numel))
