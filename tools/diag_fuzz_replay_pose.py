"""tools/diag_fuzz_replay_pose.py <seed>... -- replay fuzz draws on the GPU and put the POSE-gradient errors (max abs against the float64
oracle, relative to its max) of every backward owner and of the oracle's own fp32 build side by side, with the rays masked as fragile.
(Diagnostic; the oracle is the checker here as in the tests.)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_parity as T
from helpers import FragileRays, edge_jump_for
from oracle.mvp_oracle import Oracle
import ava256_amd as ops
if os.environ.get("MVP_VARIANT_LIB"):   # a build_variants/ library instead of the product (diagnosis only)
    from ava256_amd import _lib
    _lib.use_library(os.path.abspath(os.environ["MVP_VARIANT_LIB"]))
o64, o32 = Oracle("f64"), Oracle("f32")
for seed in (int(x) for x in sys.argv[1:]):
    c = T.fuzz_draw(seed, o64)
    a, fs, fe, warp = c["args"], c["fadescale"], c["fadeexp"], c["warp"]
    ref_rgba, ref_sat, st = o64.march_forward(*a, fadescale=fs, fadeexp=fe, ray_diagnostics=True, warp=warp)
    print("==", c["cfg"], "rays hit", st["rays_hit"], "saturated", st.get("rays_saturated"))
    for mode in T.BACKWARD_MODES:
        fragile = FragileRays(ref_sat, st["margin"], c["gout"], nsamples=st["nsamples"], max_frac=0.01, min_allowed=3, edge=st["edge"],
                              edge_jump=edge_jump_for(T.FWD_TOL * max(1.0, np.abs(ref_rgba).max()), a[7]))
        rgba, grads, diag = T._march(ops, *a, fs, fe, grad_out=fragile, mode=mode, warp=warp)
        g2 = fragile.masked()
        ref = o64.march_backward(*a, ref_sat, g2, fadescale=fs, fadeexp=fe, warp=warp)
        r32 = o32.march_backward(*a, ref_sat, g2, fadescale=fs, fadeexp=fe, warp=warp)
        out = []
        for i, k in enumerate(("primpos", "primrot", "primscale", "template")):
            m = np.abs(ref[i]).max()
            out.append("%s kernel %.3e fp32-oracle %.3e (max %.3e)" % (k, np.abs(grads[k] - ref[i]).max() / m, np.abs(r32[i] - ref[i]).max() / m, m))
        print("  ", mode, "masked rays", int(fragile.mask.sum()), "|", " | ".join(out))
