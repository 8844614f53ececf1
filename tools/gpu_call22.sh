#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02v; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"
tail -5 $O/pytest.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train"
timeout 300 $B > $O/bench.json 2>> $O/bench.err
timeout 300 $B --alpha-gain 20 > $O/bench_a20.json 2>> $O/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02v/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f" % d["ms_per_step"], d["kernel_ms"])
    except Exception as e: print(f, "ERR", e)
PY
