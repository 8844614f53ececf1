#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02y; mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc $?"
tail -3 $O/bench_full.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02y/bench_full.json").read().strip().splitlines()[-1])
print("ms/step %.2f" % d["ms_per_step"], d["kernel_ms"])
for k,v in d["train"].items():
    if isinstance(v, dict): print(k, "%.1f it/s %.2f ms" % (v["iters_per_s"], v["ms_per_iter"]), v["kernel_ms"], v["final_loss"], v["background_mlp"][:40])
PY
