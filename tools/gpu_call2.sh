#!/bin/bash
# round-2 call 2: GPU suite on the lane-independent forward + hardened backward, then A/B benches of the sweep variants
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02b; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -4 $O/pytest.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline"
timeout 300 $B > $O/bench_m32.json 2> $O/bench_m32.err; cut -c1-600 $O/bench_m32.json
timeout 300 $B --workload C3 > $O/bench_m32_C3.json 2>> $O/bench_m32.err
timeout 300 $B --workload C4 > $O/bench_m32_C4.json 2>> $O/bench_m32.err
timeout 300 $B --alpha-gain 20 > $O/bench_m32_a20.json 2>> $O/bench_m32.err
cp ava-256_amd/libmvp_gfx950.so /tmp/prod.so
cp build_variants/libmvp_m24.so ava-256_amd/libmvp_gfx950.so
timeout 300 $B > $O/bench_m24.json 2> $O/bench_m24.err
timeout 300 $B --workload C3 > $O/bench_m24_C3.json 2>> $O/bench_m24.err
cp build_variants/libmvp_dbg.so ava-256_amd/libmvp_gfx950.so
MVP_DEBUG_SLOT_SWEEP=1 timeout 300 $B > $O/bench_slot.json 2> $O/bench_slot.err
MVP_DEBUG_SLOT_SWEEP=1 timeout 300 $B --workload C3 > $O/bench_slot_C3.json 2>> $O/bench_slot.err
cp /tmp/prod.so ava-256_amd/libmvp_gfx950.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02b/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f" % d["ms_per_step"], {k:round(v,3) for k,v in d["kernel_ms"].items()})
    except Exception as e: print(f, "ERR", e)
PY
