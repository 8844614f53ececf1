#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02w; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train"
cp ava-256_amd/libmvp_gfx950.so /tmp/prod.so
for rep in 1 2; do
for v in prev prod; do
  if [ $v = prev ]; then cp build_variants/libmvp_prev.so ava-256_amd/libmvp_gfx950.so; else cp /tmp/prod.so ava-256_amd/libmvp_gfx950.so; fi
  timeout 300 $B > $O/bench_${v}_$rep.json 2>> $O/bench.err
  timeout 300 $B --alpha-gain 20 > $O/bench_${v}_a20_$rep.json 2>> $O/bench.err
done; done
cp /tmp/prod.so ava-256_amd/libmvp_gfx950.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02w/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f" % d["ms_per_step"], {k: round(v,3) for k,v in d["kernel_ms"].items()})
    except Exception as e: print(f, "ERR", e)
PY
