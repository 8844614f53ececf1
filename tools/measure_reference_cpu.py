#!/usr/bin/env python3
"""tools/measure_reference_cpu.py -- time the REFERENCE's own CPU-runnable statement of the raymarch in this build container
and record it, stamped, in profiles/reference_cpu.json (bench.py carries it as `reference_cpu`).

What is timed: the dense pure-PyTorch raymarch + autograd backward of /root/reference/extensions/mvpraymarch/mvpraymarch.py
(`gradcheck`, lines 553-641) on its STOCK scene (N=2, 65x65 rays, K=64 primitives, 32^3 slabs, fp32), executed from the
mounted file with "cuda" redirected to "cpu" -- the reference prints its own forward / backward seconds ("pytime"), which is
what is parsed.  The reference's Python cannot travel to the GPU box, so this is a build-container number (core count
recorded), carried beside the live `cpu_baseline` that bench.py measures on the GPU box's host.
The `Autoencoder` forward / backward timing of the survey is carried over unchanged, with its source.
Runs only where /root/reference is mounted.  ~30 s."""
import contextlib
import io
import json
import os
import re
import subprocess
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def dense_oracle_seconds():
    sys.modules["mvpraymarchlib"] = types.ModuleType("mvpraymarchlib")  # CUDA module stand-in
    src = open(os.path.join(REF, "extensions/mvpraymarch/mvpraymarch.py")).read()
    for a, b in (('"cuda"', '"cpu"'), ("torch.cuda.synchronize()", "pass"), ("from . import mvpraymarchlib", "import mvpraymarchlib")):
        assert a in src, a
        src = src.replace(a, b)
    ns = {"__name__": "refmvp"}
    exec(compile(src, "refmvp", "exec"), ns)

    class Done(Exception):
        pass

    def stop(*a, **k):   # the CUDA entry point: the dense statement has run by the time gradcheck calls it
        raise Done()

    ns["mvpraymarch"] = stop
    buf = io.StringIO()
    t0 = time.time()
    with contextlib.redirect_stdout(buf):
        try:
            ns["gradcheck"](usebvh="fixedorder", sortprims=False, maxhitboxes=512, synchitboxes=True, dowarp=False,
                            chlast=True, fadescale=8.0, fadeexp=8.0, accum=0, algo=0, griddim=3)
        except Done:
            pass
    total = time.time() - t0
    m = re.search(r"pytime\s+([0-9.eE+-]+)\s+([0-9.eE+-]+)\s+([0-9.eE+-]+)", buf.getvalue())
    assert m, buf.getvalue()[-500:]
    return float(m.group(1)), float(m.group(2)), total


if __name__ == "__main__":
    fwd, bwd, total = dense_oracle_seconds()
    rays = 2 * 65 * 65
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    out = {
        "_measured_at_commit": commit, "_measured_on": time.strftime("%Y-%m-%d"), "_script": "tools/measure_reference_cpu.py",
        "where": "build container, %d CPU cores, torch %s on CPU; NOT the GPU box's host" % (os.cpu_count(), torch.__version__),
        "dense_raymarch_oracle": {"scene": "mvpraymarch.py gradcheck stock scene: N=2, 65x65, K=64, 32^3 slabs, fp32",
                                  "source": "reference's own 'pytime' line (mvpraymarch.py:636-637), executed from the mounted file",
                                  "fwd_s": fwd, "bwd_s": bwd, "fwd_rays_per_s": rays / fwd, "fwd_bwd_rays_per_s": rays / (fwd + bwd),
                                  "wall_s_of_the_run": total},
        "autoencoder": {"model": "reference Autoencoder, K=16384, 46.9 M params, batch 1, 128x128, raymarch stubbed",
                        "fwd_s": 4.6, "bwd_s": 1.5, "source": "BASELINE.md section 2 (survey commit 02e6dcd, 8 cores); not re-measured"},
    }
    path = os.path.join(ROOT, "profiles", "reference_cpu.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["dense_raymarch_oracle"]))
