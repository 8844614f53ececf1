#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02t; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train"
cp ava-256_amd/libmvp_gfx950.so /tmp/prod.so
cp build_variants/libmvp_dbg.so ava-256_amd/libmvp_gfx950.so
for st in 11 13 1 2 0; do
  MVP_DEBUG_STAGE=$st timeout 300 $B > $O/bench_stage$st.json 2>> $O/bench.err
done
cp /tmp/prod.so ava-256_amd/libmvp_gfx950.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02t/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "fwd %.3f" % d["kernel_ms"]["march_forward"])
    except Exception as e: print(f, "ERR", e)
PY
