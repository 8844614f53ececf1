"""Bandwidth of the decoder -> raymarch hand-off kernel vs the eager PyTorch expression it replaces (row N2)."""
import json, sys, torch
sys.path.insert(0, "/root/repo")
from ava256_amd.assemble import assemble_template
N, nh, B = 4, 128, 8          # ava-256's shipped size: 16384 primitives, 8^3 slabs, batch 4 per GPU
S = nh * B
tex = torch.randn(N, 3 * B, S, S, device="cuda", requires_grad=True)
op = torch.randn(N, B, S, S, device="cuda", requires_grad=True)
gout = torch.randn(N, nh * nh, B, B, B, 4, device="cuda")
def eager():
    rgb = tex.view(N, B, 3, nh, B, nh, B).permute(0, 3, 5, 1, 4, 6, 2).reshape(N, nh * nh, B, B, B, 3)
    a = op.view(N, B, 1, nh, B, nh, B).permute(0, 3, 5, 1, 4, 6, 2).reshape(N, nh * nh, B, B, B, 1)
    return torch.cat([torch.relu(rgb * 25.0 + 100.0), torch.relu(a)], dim=-1)
def timeit(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
def f_hip():
    return assemble_template(tex, op, nh * nh, B)
def fb(fn):
    tex.grad = None; op.grad = None
    fn().backward(gout)
fwd_bytes = 2 * N * 4 * B * S * S * 4           # read 4B planes, write the same number of floats
bwd_bytes = 3 * N * 4 * B * S * S * 4           # read tplate + grad_tplate, write grads
t_f, t_e = timeit(f_hip), timeit(eager)
t_fb, t_eb = timeit(lambda: fb(f_hip)), timeit(lambda: fb(eager))
print(json.dumps({"shape": [N, nh, B], "hip_fwd_ms": t_f, "hip_fwd_GBps": fwd_bytes / t_f / 1e6, "eager_fwd_ms": t_e,
                  "hip_fwd_bwd_ms": t_fb, "hip_bwd_GBps": bwd_bytes / max(t_fb - t_f, 1e-9) / 1e6, "eager_fwd_bwd_ms": t_eb}))
