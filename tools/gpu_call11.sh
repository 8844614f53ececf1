#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02k; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train"
cp ava-256_amd/libmvp_gfx950.so /tmp/prod.so
for v in l8k_a l8k_b l8k_c; do
  cp build_variants/libmvp_$v.so ava-256_amd/libmvp_gfx950.so
  timeout 300 $B > $O/bench_$v.json 2> $O/bench_$v.err
  timeout 300 $B --workload C3 > $O/bench_${v}_C3.json 2>> $O/bench_$v.err
  timeout 300 $B --workload C4 > $O/bench_${v}_C4.json 2>> $O/bench_$v.err
  timeout 300 $B --alpha-gain 20 > $O/bench_${v}_a20.json 2>> $O/bench_$v.err
done
cp /tmp/prod.so ava-256_amd/libmvp_gfx950.so
timeout 300 $B > $O/bench_prod.json 2> $O/bench_prod.err
timeout 600 python -m pytest tests/test_gpu_hardening.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02k/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f" % d["ms_per_step"], {k:round(v,3) for k,v in d.get("kernel_ms",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
