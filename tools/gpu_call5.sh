#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02q; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train"
cp ava-256_amd/libmvp_gfx950.so /tmp/prod.so
for v in exp1 exp2 exp3; do
  cp build_variants/libmvp_$v.so ava-256_amd/libmvp_gfx950.so
  timeout 300 $B > $O/bench_$v.json 2> $O/bench_$v.err
  timeout 300 $B --workload C3 > $O/bench_${v}_C3.json 2>> $O/bench_$v.err
done
cp /tmp/prod.so ava-256_amd/libmvp_gfx950.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02q/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f" % d["ms_per_step"], {k:round(v,3) for k,v in d.get("kernel_ms",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
