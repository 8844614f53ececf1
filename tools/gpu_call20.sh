#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02s; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train"
cp ava-256_amd/libmvp_gfx950.so /tmp/prod.so
for v in rotB rotC rotD pad3 pad9 e6; do
  cp build_variants/libmvp_$v.so ava-256_amd/libmvp_gfx950.so
  for w in C2 C4; do timeout 300 $B --workload $w > $O/bench_${v}_$w.json 2>> $O/bench.err; done
  timeout 300 $B --alpha-gain 20 > $O/bench_${v}_a20.json 2>> $O/bench.err
done
cp /tmp/prod.so ava-256_amd/libmvp_gfx950.so
for w in C2 C4; do timeout 300 $B --workload $w > $O/bench_prod_$w.json 2>> $O/bench.err; done
timeout 300 $B --alpha-gain 20 > $O/bench_prod_a20.json 2>> $O/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02s/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "bwd %.3f" % d["kernel_ms"]["march_backward"])
    except Exception as e: print(f, "ERR", e)
PY
