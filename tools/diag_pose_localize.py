"""tools/diag_pose_localize.py <seed> -- where does a fuzz draw's pose-gradient error come from?  Finds the worst primpos entry of the
primitive-centric backward against the float64 oracle, then bisects over the RAYS (gradients are linear in the upstream gradient: one
forward + backward per half with the other rays' upstream gradient zeroed) down to the ray that carries the error, and prints what the
oracle knows about it.  (Diagnostic; the oracle is the checker here as in the tests.)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_parity as T
from helpers import FragileRays, edge_jump_for
from oracle.mvp_oracle import Oracle
import ava256_amd as ops
o64 = Oracle("f64")
seed = int(sys.argv[1])
WHAT = sys.argv[2] if len(sys.argv) > 2 else "primpos"   # or "template"
IDX = {"primpos": 0, "primrot": 1, "primscale": 2, "template": 3}[WHAT]
c = T.fuzz_draw(seed, o64)
a, fs, fe, warp = c["args"], c["fadescale"], c["fadeexp"], c["warp"]
ref_rgba, ref_sat, st = o64.march_forward(*a, fadescale=fs, fadeexp=fe, ray_diagnostics=True, warp=warp)
fragile = FragileRays(ref_sat, st["margin"], c["gout"], nsamples=st["nsamples"], max_frac=0.01, min_allowed=3, edge=st["edge"],
                      edge_jump=edge_jump_for(T.FWD_TOL * max(1.0, np.abs(ref_rgba).max()), a[7]))
rgba, grads, diag = T._march(ops, *a, fs, fe, grad_out=fragile, mode="prim", warp=warp)
g_all = fragile.masked()
ref = o64.march_backward(*a, ref_sat, g_all, fadescale=fs, fadeexp=fe, warp=warp)
err = grads[WHAT] - ref[IDX]
w = np.unravel_index(np.abs(err).argmax(), err.shape)
print(c["cfg"]); print("worst", WHAT, "entry", w, "kernel %.6e oracle %.6e error %.4e (max |g| %.4e)" % (grads[WHAT][w], ref[IDX][w], err[w], np.abs(ref[IDX]).max()))
n = w[0]


def error_of(gout, mode="prim"):
    _, g, _ = T._march(ops, *a, fs, fe, grad_out=gout, mode=mode, warp=warp)
    r = o64.march_backward(*a, ref_sat, gout, fadescale=fs, fadeexp=fe, warp=warp)
    return g[WHAT][w] - r[IDX][w], g, r


live = np.argwhere(np.abs(g_all[n]).max(-1) > 0)          # rays of image n with an upstream gradient
print("rays with gradient in image", n, ":", len(live))
cand = live
while len(cand) > 1:
    half = len(cand) // 2
    best = None
    for part in (cand[:half], cand[half:]):
        g = np.zeros_like(g_all)
        for (y, x) in part:
            g[n, y, x] = g_all[n, y, x]
        e, _, _ = error_of(g)
        if best is None or abs(e) > abs(best[0]):
            best = (e, part)
    print("  %5d rays -> error %.4e" % (len(best[1]), best[0]))
    cand = best[1]
y, x = cand[0]
g = np.zeros_like(g_all); g[n, y, x] = g_all[n, y, x]
for mode in ("prim", "ray"):
    e, gk, r = error_of(g, mode)
    print("ray (%d, %d, %d) alone, %s: %s[w] kernel %.6e oracle %.6e error %.4e | everywhere: max |error| %.4e at %s" % (
        n, y, x, mode, WHAT, gk[WHAT][w], r[IDX][w], e, np.abs(gk[WHAT] - r[IDX]).max(),
        np.unravel_index(np.abs(gk[WHAT] - r[IDX]).argmax(), r[IDX].shape)))
print("oracle: margin %.3e edge %.3e nsamples %d ref_sat %s rgba %s gout %s" % (st["margin"][n, y, x], st["edge"][n, y, x], st["nsamples"][n, y, x],
                                                                               ref_sat[n, y, x], ref_rgba[n, y, x], g_all[n, y, x]))
rgba_k, _, _ = T._march(ops, *a, fs, fe, grad_out=None, mode="prim", warp=warp)
print("kernel rgba", rgba_k[n, y, x], "ray o", a[0][n, y, x], "d", a[1][n, y, x], "tminmax", a[3][n, y, x], "dt", a[2])
