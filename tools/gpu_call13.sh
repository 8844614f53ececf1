#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02l; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train"
cp ava-256_amd/libmvp_gfx950.so /tmp/prod.so
cp build_variants/libmvp_dbg.so ava-256_amd/libmvp_gfx950.so
for pad in 0 14000 30000; do
  MVP_DEBUG_LDS_PAD=$pad timeout 300 $B > $O/bench_pad$pad.json 2> $O/bench_pad$pad.err
  MVP_DEBUG_LDS_PAD=$pad timeout 300 $B --workload C3 > $O/bench_pad${pad}_C3.json 2>> $O/bench_pad$pad.err
done
cp /tmp/prod.so ava-256_amd/libmvp_gfx950.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02l/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f" % d["ms_per_step"], {k:round(v,3) for k,v in d.get("kernel_ms",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
