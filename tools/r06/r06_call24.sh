#!/bin/bash
# round-6 GPU call 24: the backward's per-round scales with v_rcp instead of five IEEE divisions per wave and round: the whole GPU
# suite, A/B against the previous build at C2 / C3 / C4 / saturated.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06v; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -3 $O/tests.log
M="--steps 20 --warmup 5 --no-cpu-baseline --no-train --no-render"
for i in 1 2 3; do
  for wl in "C2" "C2 --alpha-gain 40" "C3" "C4"; do
    timeout 200 python tools/bench_variant.py build_variants/libmvp_r06rd.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('prev', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new ', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
  done
done
