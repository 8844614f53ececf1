#!/bin/bash
# round-6 GPU call 36: backward rounds bounded by the rays their records NAME (shared compaction, up to 7 entries per wave in one
# round): the GPU suite, then product against the previous build (build_variants/libmvp_prev.so) on the bench scene (C2, saturated,
# C3, C4) and in the C2 train leg.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ac; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -4 $O/tests.log
M="--steps 20 --warmup 5 --no-cpu-baseline --no-train --no-render --no-workloads"
T='import sys, json
from ava256_amd import _lib
if sys.argv[1] != "product": _lib.use_library(sys.argv[1])
import bench
bench.main(["--mode", "train", "--workload", "C2", "--steps", "12", "--warmup", "4", "--bg", "off"])'
for i in 1 2 3; do
  for wl in "C2" "C2 --alpha-gain 40" "C3" "C4"; do
    timeout 200 python tools/bench_variant.py build_variants/libmvp_prev.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('prev', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new ', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
  done
  timeout 300 python -c "$T" build_variants/libmvp_prev.so 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); t=d.get('train', d); t=t.get('C2', t); print('prev train', t.get('iters_per_s'), t.get('kernel_ms'))" | tee -a $O/ab.txt
  timeout 300 python -c "$T" product 2>/dev/null < /dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); t=d.get('train', d); t=t.get('C2', t); print('new  train', t.get('iters_per_s'), t.get('kernel_ms'))" | tee -a $O/ab.txt
done
