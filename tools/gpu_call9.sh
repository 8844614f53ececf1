#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02i; mkdir -p $O
timeout 900 python -m pytest tests/test_native_shim.py tests/test_trainloop.py tests/test_placement.py tests/test_gradclip.py -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02i/bench_default.json").read().strip().splitlines()[-1])
print("ms/step %.2f value %.4g" % (d["ms_per_step"], d["value"]), {k:round(v,3) for k,v in d["kernel_ms"].items()})
print("roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"],4), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
for k,v in d["train"].items():
    if isinstance(v,dict): print("train",k, round(v["iters_per_s"],1), "it/s", round(v["ms_per_iter"],2), "ms", {a:round(b,2) for a,b in v["kernel_ms"].items()})
PY
