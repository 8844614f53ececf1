"""tools/diag_fuzz_replay_forward.py <seed>... -- replay the FORWARD of fuzz draws (tests/test_gpu_parity.py::test_randomized_configurations) on
the GPU and list the rays within half the tolerance of failing, with the float64 oracle's diagnostics (edge, margin, samples) and the
error of the oracle's own fp32 build beside the kernel's.  (Diagnostic; the oracle is the checker here as in the tests.)"""
import os, sys
import numpy as np
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_parity as T
from helpers import FragileRays, EDGE_JUMP, SAT_BAND, edge_jump_for
from oracle.mvp_oracle import Oracle
import ava256_amd as ops
o64, o32 = Oracle("f64"), Oracle("f32")
for seed in (int(x) for x in sys.argv[1:]):
    c = T.fuzz_draw(seed, o64)
    a, fs, fe, warp = c["args"], c["fadescale"], c["fadeexp"], c["warp"]
    ref_rgba, ref_sat, st = o64.march_forward(*a, fadescale=fs, fadeexp=fe, ray_diagnostics=True, warp=warp)
    r32, s32, _ = o32.march_forward(*a, fadescale=fs, fadeexp=fe, ray_diagnostics=True, warp=warp)
    tol0 = T.FWD_TOL * max(1.0, np.abs(ref_rgba).max())
    ej = edge_jump_for(tol0, a[7])
    fragile = FragileRays(ref_sat, st["margin"], c["gout"], nsamples=st["nsamples"], max_frac=0.01, min_allowed=3, edge=st["edge"], edge_jump=ej)
    rgba, grads, diag = T._march(ops, *a, fs, fe, grad_out=fragile, mode="prim", warp=warp)
    err = np.abs(rgba - ref_rgba).max(-1)
    e32 = np.abs(r32 - ref_rgba).max(-1)
    tol = T.FWD_TOL * max(1.0, np.abs(ref_rgba).max())
    bad = (err > tol) & ~fragile.mask
    print(c["cfg"], "tol", tol, "bad rays", int(bad.sum()), "EDGE_JUMP", EDGE_JUMP, "edge_jump_for", ej)
    for idx in zip(*np.nonzero(err > 0.5 * tol)):
        print("  ray", idx, "err %.5f fp32-oracle err %.5f masked %s edge %.3e margin %.3e nsamples %d rgba_ref %s" % (
            err[idx], e32[idx], bool(fragile.mask[idx]), st["edge"][idx], st["margin"][idx], st["nsamples"][idx], np.round(ref_rgba[idx], 3)))
