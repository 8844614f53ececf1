#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -5 $O/pytest.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train"
timeout 300 $B > $O/bench_pipe.json 2> $O/bench_pipe.err
timeout 300 $B --workload C3 > $O/bench_pipe_C3.json 2>> $O/bench_pipe.err
timeout 300 $B --workload C4 > $O/bench_pipe_C4.json 2>> $O/bench_pipe.err
timeout 300 $B --alpha-gain 20 > $O/bench_pipe_a20.json 2>> $O/bench_pipe.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02f/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], "ms/step %.2f" % d["ms_per_step"], {k:round(v,3) for k,v in d.get("kernel_ms",{}).items()})
    except Exception as e: print(f, "ERR", e)
PY
bash tools/pmc.sh r02f_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" --steps 3 --warmup 1 --no-cpu-baseline --no-train | grep "false, true, false, 8"
bash tools/pmc.sh r02f_ta "TA_TA_BUSY_sum TA_BUSY_max TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" --steps 3 --warmup 1 --no-cpu-baseline --no-train | grep "false, true, false, 8"
