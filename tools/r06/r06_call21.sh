#!/bin/bash
# round-6 GPU call 21: the MLP parity tests with the third (ragged) reference-made fixture; what they measure is written to
# gpurun_out/bgmlp_parity.json.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06t; mkdir -p $O
timeout 300 python -m pytest tests/test_bgmlp.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -3 $O/tests.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/bgmlp_parity.json"))
for f, rows in d.items():
    worst_c = min(v["cosine"] for k, v in rows.items() if isinstance(v, dict))
    worst_n = max(v["norm_wise"] for k, v in rows.items() if isinstance(v, dict))
    print(f, "output %.4f of the spread; worst cosine %.4f, worst norm-wise %.3f" % (rows["output_max_abs_over_spread"], worst_c, worst_n))
PY
