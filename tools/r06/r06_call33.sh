#!/bin/bash
# round-6 GPU call 33: XCDs per shared image (F) re-swept under the two-run strip order: R = 4 images with F = 2 (product) / 4 / 8
# at C3 and C4; R = 2 with F = 4 (product) / 8.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ab; mkdir -p $O
M="--steps 30 --warmup 5 --no-cpu-baseline --no-train --no-render"
for i in 1 2; do
  for wl in "C3" "C4"; do
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('F2  ', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/fsweep.txt
    for v in f4 f8; do
    timeout 200 python tools/bench_variant.py build_variants/libmvp_$v.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v  ', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/fsweep.txt
    done
  done
  timeout 200 python bench.py $M --workload C3 --cams 2 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('R2F4', 'C3x2', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/fsweep.txt
  timeout 200 python tools/bench_variant.py build_variants/libmvp_r2f8.so $M --workload C3 --cams 2 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('R2F8', 'C3x2', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/fsweep.txt
done
