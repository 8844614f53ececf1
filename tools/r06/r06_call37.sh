#!/bin/bash
# round-6 GPU call 37: strip height (packet rows per dispatch strip) re-swept under the two-run order for shared images:
# 1 / 2 / 3 (product) / 4 / 6 rows at C3, C4 (and C2 as a control).
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ab; mkdir -p $O
M="--steps 30 --warmup 5 --no-cpu-baseline --no-train --no-render --no-workloads"
for i in 1 2; do
  for wl in "C3" "C4" "C2"; do
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('rows3', '$wl', d['ms_per_step'], d['kernel_ms']['march_forward'])" | tee -a $O/striprows.txt
    for r in 1 2 4 6; do
      timeout 200 python tools/bench_variant.py build_variants/libmvp_sr$r.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('rows$r', '$wl', d['ms_per_step'], d['kernel_ms']['march_forward'])" | tee -a $O/striprows.txt
    done
  done
done
