#!/bin/bash
# round-6 GPU call 22: the final tree as the driver will run it: build check of what is loaded, smoke, the whole GPU suite, the default bench line.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06final; mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 < /dev/null | tail -1
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -3 $O/tests.log
T0=$(date +%s); timeout 900 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null; echo "bench rc $? in $(( $(date +%s) - T0 )) s"
python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["kernel_ms"])
print("saturated", d["saturated"]["ms_per_step"], "train_like", d["train_like"]["ms_per_step"], d["train_like"]["backward_marks"])
for k, v in d["train"].items():
    if isinstance(v, dict): print(k, v["iters_per_s"], v["steps"])
PY
