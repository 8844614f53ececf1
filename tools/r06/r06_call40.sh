#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ab; mkdir -p $O
M="--steps 30 --warmup 5 --no-cpu-baseline --no-train --no-render --no-workloads"
for i in 1 2 3; do
  for wl in "C2" "C3" "C4"; do
    timeout 200 python tools/bench_variant.py build_variants/libmvp_prevdiv.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('prev', '$wl', d['ms_per_step'], d['kernel_ms']['march_forward'], d['kernel_ms']['march_backward'])" | tee -a $O/fastdiv.txt
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new ', '$wl', d['ms_per_step'], d['kernel_ms']['march_forward'], d['kernel_ms']['march_backward'])" | tee -a $O/fastdiv.txt
  done
done
