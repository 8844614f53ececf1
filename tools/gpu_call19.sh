#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02r; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train"
for w in C2 C3 C4; do timeout 300 $B --workload $w > $O/bench_rot_$w.json 2>> $O/bench.err; done
timeout 300 $B --alpha-gain 20 > $O/bench_rot_a20.json 2>> $O/bench.err
cp ava-256_amd/libmvp_gfx950.so /tmp/prod.so
cp build_variants/libmvp_norot.so ava-256_amd/libmvp_gfx950.so
for w in C2 C3 C4; do timeout 300 $B --workload $w > $O/bench_norot_$w.json 2>> $O/bench.err; done
timeout 300 $B --alpha-gain 20 > $O/bench_norot_a20.json 2>> $O/bench.err
cp /tmp/prod.so ava-256_amd/libmvp_gfx950.so
bash tools/pmc.sh r02r_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" --steps 3 --warmup 1 --no-cpu-baseline --no-train | grep bwd_prim
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02r/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f" % d["ms_per_step"], {k:round(v,3) for k,v in d.get("kernel_ms",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
