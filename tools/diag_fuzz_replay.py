"""tools/diag_fuzz_replay.py <seed>... -- replay draws of tests/test_gpu_parity.py::test_randomized_configurations on the GPU
and, per primitive, put the slab-gradient error of every backward owner (primitive-centric, ray-centric, capacity 4)
and of the fp32 oracle side by side against the float64 oracle.  (Diagnostic; runs where tests/ and oracle/ are: the oracle is the checker here as in the tests.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_parity as T  # noqa: E402
from helpers import FragileRays  # noqa: E402
from oracle.mvp_oracle import Oracle  # noqa: E402
import ava256_amd as ops  # noqa: E402

o64, o32 = Oracle("f64"), Oracle("f32")
for seed in (int(x) for x in sys.argv[1:]):
    c = T.fuzz_draw(seed, o64)
    a, fs, fe, warp, N, K = c["args"], c["fadescale"], c["fadeexp"], c["warp"], c["N"], c["K"]
    ref_rgba, ref_sat, st = o64.march_forward(*a, fadescale=fs, fadeexp=fe, ray_diagnostics=True, warp=warp)
    print("==", c["cfg"], "rays hit", st["rays_hit"])
    res = {}
    fragile = FragileRays(ref_sat, st["margin"], c["gout"], max_frac=0.05, min_allowed=50, edge=st["edge"])
    print("   edge-fragile rays: %d of %d hit" % (int(fragile.edge_mask.sum()), st["rays_hit"]))
    for mode in T.BACKWARD_MODES:
        try:
            rgba, grads, diag = T._march(ops, *a, fs, fe, grad_out=fragile, mode=mode, warp=warp)
        except AssertionError as e:
            print("   ", mode, "fragile-ray check:", str(e)[:200])
            continue
        res[mode] = (grads, diag, fragile.mask.copy())
    if not res:
        continue
    mask = next(iter(res.values()))[2]
    g2 = c["gout"].copy()
    g2[mask] = 0.0
    ref = o64.march_backward(*a, ref_sat, g2, fadescale=fs, fadeexp=fe, warp=warp)
    r32_rgba, r32_sat, _ = o32.march_forward(*a, fadescale=fs, fadeexp=fe, ray_diagnostics=True, warp=warp)
    ref32 = o32.march_backward(*a, r32_sat, g2, fadescale=fs, fadeexp=fe, warp=warp)
    refk = ref[3].reshape(N * K, -1)
    pmax = np.abs(refk).max(1)
    live = pmax > 0
    rows = {"oracle32": np.abs(ref32[3].reshape(N * K, -1) - refk).max(1)}
    for mode, (grads, diag, m) in res.items():
        assert (m == mask).all()
        rows[mode] = np.abs(grads["template"].reshape(N * K, -1) - refk).max(1)
        print("   %-5s flags %#x two-pass %d handed-over %d" % (mode, diag.get("handoff_flags", 0), diag.get("prims_two_pass", -1),
                                                               diag.get("prims_handed_over", -1)))
    for name, e in rows.items():
        rel = np.where(live, e / np.maximum(pmax, 1e-300), 0.0)
        w = int(rel.argmax())
        print("   %-9s per-primitive rel err max %.3e (prim %d: pmax %.3e, global max %.3e) median %.1e  dead-prim max abs %.2e" % (
            name, rel[w], w, pmax[w], pmax.max(), np.median(rel[live]), e[~live].max() if (~live).any() else 0.0))
    # the fixed-point statement of the primitive-centric kernel: error against the A-PRIORI bound of a
    # round's values, B_rgb = G * min(1, Amax_k * dt), B_a = G * dt * (3 (Tmax_k + Rmax) + 1)
    tpl = a[7].reshape(N * K, -1, 4)
    dt, G = float(a[2]), np.abs(g2).max()
    Brgb = G * np.minimum(1.0, np.abs(tpl[..., 3]).max(1) * dt)
    Ba = G * dt * (3.0 * (np.abs(tpl[..., :3]).max((1, 2)) + np.abs(tpl[..., :3]).max()) + 1.0)
    for mode, (grads, diag, m) in res.items():
        e = np.abs(grads["template"].reshape(N * K, -1, 4) - ref[3].reshape(N * K, -1, 4))
        print("   %-5s err / a-priori bound: rgb max %.2e  alpha max %.2e" % (mode, (e[..., :3].max((1, 2)) / Brgb).max(),
                                                                           (e[..., 3].max(1) / Ba).max()))
    w = int(np.where(live, rows.get("prim", rows[next(iter(rows))]) / np.maximum(pmax, 1e-300), 0).argmax())
    print("   worst primitive of 'prim': %d; its errors:" % w, {k: "%.3e" % v[w] for k, v in rows.items()}, "pmax %.3e" % pmax[w])
