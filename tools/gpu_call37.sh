#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02y; mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline --no-train > $O/bench_render.json 2> $O/bench_render.err; echo "rc $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02y/bench_render.json").read().strip().splitlines()[-1])
print("ms/step %.2f" % d["ms_per_step"], d["kernel_ms"], d.get("render"))
PY
