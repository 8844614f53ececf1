"""Per-ray / per-packet statistics of the synthetic scenes (uses the CPU oracle: analysis tooling, not product code).
Prints what the forward sweep design needs: primitives listed per ray, samples per ray, and for 8x8 packets the lane
utilisation a lane-independent sweep would reach (sum of samples / (64 * max samples in the packet))."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ava256_amd.scene import make_scene
from oracle.mvp_oracle import Oracle

def main():
    o = Oracle("f32")
    for name, H, W, K, gain in (("C2", 512, 512, 4096, 1.0), ("C2a20", 512, 512, 4096, 20.0), ("C3", 512, 512, 16384, 1.0), ("C4", 1024, 1024, 8192, 1.0)):
        s = make_scene(1, H, W, K, device="cpu", seed=1112, alpha_gain=gain)
        rp, rd, tm = o.raydirs(s["campos"].numpy(), s["camrot"].numpy(), s["focal"].numpy(), s["princpt"].numpy(), s["pixelcoords"].numpy(), s["volradius"])
        rgba, sat, st = o.march_forward(rp, rd, s["stepsize"], tm, s["primpos"].numpy(), s["primrot"].numpy(), s["primscale"].numpy(), s["template"].numpy(), ray_diagnostics=True)
        hc, ns = st["hitcount"][0], st["nsamples"][0]
        hit = hc > 0
        print(name, "rays hit %.3f" % hit.mean(), "hitcount mean %.1f p99 %d max %d" % (hc[hit].mean(), np.percentile(hc[hit], 99), hc.max()),
              "samples mean %.1f p99 %d max %d" % (ns[hit].mean(), np.percentile(ns[hit], 99), ns.max()))
        # packets
        hp = hc.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
        sp = ns.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
        live = hp.max(1) > 0
        hp, sp = hp[live], sp[live]
        mx = sp.max(1)
        print("   packets hit %d; lane-independent utilisation: sum samples / (64 * sum of packet max) = %.3f ; mean packet max samples %.1f" % (
            live.sum(), sp.sum() / (64.0 * mx.sum()), mx.mean()))
        for M in (12, 16, 20, 24, 31):
            print("   max crossings per ray <= %d in %.4f of the hit packets" % (M, (hp.max(1) <= M).mean()), end=";")
        print()

main()
