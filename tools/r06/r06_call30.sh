#!/bin/bash
# round-6 GPU calls 30 / 31: forward dispatch order when few images share the chip (N % 8 != 0).  An XCD walks ITS strips of an
# image (a) centre-out (heavy first), (b) in two interleaved runs (top half / bottom half: heavy and light strips mixed in time)
# instead of top-down -- A/B at C3 / C4 (and C2, where nothing may change).  usage: r06_call30.sh <variant>
set -u
V=${1:-centre}
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ab; mkdir -p $O
M="--steps 30 --warmup 5 --no-cpu-baseline --no-train --no-render"
for i in 1 2 3; do
  for wl in "C3" "C4" "C2"; do
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('prod  ', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/$V.txt
    timeout 200 python tools/bench_variant.py build_variants/libmvp_$V.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$V', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/$V.txt
  done
done
