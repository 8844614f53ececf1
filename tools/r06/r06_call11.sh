#!/bin/bash
# round-6 GPU call 11: what do the forward's FALSE candidates (BVH candidates no ray of the packet crosses: ~14 of 32 per hit
# packet) cost?  Timing build in which such a candidate pays for its exact test twice (profiles/r06_fwd_false_candidates_twice.patch):
# the slowdown is an upper bound of what a tighter candidate cull could return.  Plus the diag counters (candidates / listed).
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06k; mkdir -p $O
M="--steps 20 --warmup 5 --no-cpu-baseline --no-train --no-render"
for i in 1 2 3; do
  for wl in C2 C3 C4; do
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('product', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
    timeout 200 python tools/bench_variant.py build_variants/libmvp_falsetwice.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('twice  ', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
  done
done
python - <<'PY' | tee $O/diag.txt
import torch, sys
sys.path.insert(0, ".")
import ava256_amd as ops
from ava256_amd import _hooks
from ava256_amd.scene import make_scene
for name, (N, H, W, K) in {"C2": (8, 512, 512, 4096), "C3": (4, 512, 512, 16384), "C4": (4, 1024, 1024, 8192)}.items():
    s = make_scene(N, H, W, K, device="cuda", seed=1112)
    diag = torch.zeros(8, dtype=torch.int32, device="cuda")
    _hooks.set_diag_buffer(diag)
    with torch.no_grad():
        ops.mvpraymarch_from_cameras(s["campos"], s["camrot"], s["focal"], s["princpt"], s["pixelcoords"], s["volradius"], s["stepsize"],
                                     (s["primpos"], s["primrot"], s["primscale"]), s["template"])
    torch.cuda.synchronize()
    d = _hooks.read_diag(); _hooks.set_diag_buffer(None)
    hp = max(d["packets_hit"], 1)
    print(name, d, "candidates per hit packet %.1f listed %.1f slow-path share %.3f" % (d["candidates"] / hp, d["list_entries"] / hp, d["slowpath_packets"] / hp))
PY
