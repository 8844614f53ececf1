#!/bin/bash
# round-6 GPU call 32: the two-run strip order (F > 1) in the product: the whole GPU suite, then product against the previous
# forward (build_variants/libmvp_r06rd.so) at C3 with 2 / 4 / 5 / 12 cameras and C4.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ab; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -3 $O/tests.log
M="--steps 30 --warmup 5 --no-cpu-baseline --no-train --no-render"
for i in 1 2; do
  for wl in "C3 --cams 2" "C3" "C3 --cams 5" "C3 --cams 12" "C4"; do
    timeout 200 python tools/bench_variant.py build_variants/libmvp_r06rd.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('prev', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/product.txt
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new ', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/product.txt
  done
done
