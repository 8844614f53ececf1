"""GPU microbenchmark: the background MLP's per-layer GEMM (pixels x 256) @ (256 x 256) in bf16 under the two weight
layouts -- nn.Linear's [out, in] (x @ W.T) and [in, out] (x @ W) -- forward and backward, to see which hipBLASLt kernels
torch selects (rocprofv3 --kernel-trace --stats shows the names) and what they achieve."""
import sys, time, torch
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 512 * 512
K = N = 256
dev = "cuda"
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16, requires_grad=True)
w_oi = torch.randn(N, K, device=dev, dtype=torch.bfloat16, requires_grad=True)   # nn.Linear layout
w_io = torch.randn(K, N, device=dev, dtype=torch.bfloat16, requires_grad=True)
b = torch.zeros(N, device=dev, dtype=torch.bfloat16, requires_grad=True)
gy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n


flop = 2.0 * M * K * N
for name, f in (("linear [out,in]  x @ W.T + b", lambda: torch.nn.functional.linear(x, w_oi, b)),
                ("addmm  [in,out]  x @ W + b  ", lambda: torch.addmm(b, x, w_io)),
                ("matmul [in,out]  x @ W      ", lambda: x @ w_io)):
    with torch.no_grad():
        t = timeit(f)
    print("fwd %s: %.3f ms = %.0f TFLOP/s" % (name, t, flop / t * 1e-9))

    def fb():
        x.grad = None
        y = f()
        y.backward(gy)
    t2 = timeit(fb)
    print("    fwd+bwd: %.3f ms = %.0f TFLOP/s (3 GEMMs)" % (t2, 3 * flop / t2 * 1e-9))

# the chunked weight gradient of ava-256_amd/trainloop.py (PixelLinear) against the plain path, values and time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ava256_amd.trainloop import _PixelLinearFn
xx = x.detach().clone().requires_grad_(True)
w2 = w_oi.detach().clone().requires_grad_(True)
b2 = b.detach().clone().requires_grad_(True)
y = _PixelLinearFn.apply(xx, w2, b2, 64)
y.backward(gy)
x.grad = None; w_oi.grad = None; b.grad = None
torch.nn.functional.linear(x, w_oi, b).backward(gy)
print("chunked vs plain: dW rel err %.2e, dx rel err %.2e, db rel err %.2e" % (
    float((w2.grad.float() - w_oi.grad.float()).norm() / w_oi.grad.float().norm()),
    float((xx.grad.float() - x.grad.float()).norm() / x.grad.float().norm()),
    float((b2.grad.float() - b.grad.float()).norm() / b.grad.float().norm())))


def fb2():
    xx.grad = None
    _PixelLinearFn.apply(xx, w2, b2, 64).backward(gy)


t3 = timeit(fb2)
print("PixelLinear fwd+bwd: %.3f ms = %.0f TFLOP/s (3 GEMMs)" % (t3, 3 * flop / t3 * 1e-9))
