#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -12 $O/pytest.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; cut -c1-300 $O/bench_default.json; tail -3 $O/bench_default.err
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train"
cp ava-256_amd/libmvp_gfx950.so /tmp/prod.so
for v in m32 m16; do
  cp build_variants/libmvp_$v.so ava-256_amd/libmvp_gfx950.so
  timeout 300 $B > $O/bench_$v.json 2> $O/bench_$v.err
  timeout 300 $B --workload C3 > $O/bench_${v}_C3.json 2>> $O/bench_$v.err
done
cp build_variants/libmvp_dbg.so ava-256_amd/libmvp_gfx950.so
MVP_DEBUG_SLOT_SWEEP=1 timeout 300 $B > $O/bench_slot.json 2> $O/bench_slot.err
cp /tmp/prod.so ava-256_amd/libmvp_gfx950.so
timeout 300 $B > $O/bench_m24.json 2> $O/bench_m24.err
timeout 300 $B --workload C3 > $O/bench_m24_C3.json 2>> $O/bench_m24.err
timeout 300 $B --workload C4 > $O/bench_m24_C4.json 2>> $O/bench_m24.err
timeout 300 $B --alpha-gain 20 > $O/bench_m24_a20.json 2>> $O/bench_m24.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02d/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f" % d["ms_per_step"], {k:round(v,3) for k,v in d.get("kernel_ms",{}).items()})
        if "train" in d: print("   train", {k:(round(v["iters_per_s"],1), round(v["ms_per_iter"],2), {a:round(b,2) for a,b in v["kernel_ms"].items()}) for k,v in d["train"].items() if isinstance(v,dict)})
    except Exception as e: print(f, "ERR", e)
PY
