#!/bin/bash
# round-6 GPU call 5: phase 1's list records and packet bounds as ONE batch of unconditional scalar loads (was: one dependent
# round trip per entry): backward parity tests, A/B against the mask-compaction build (the reference of call 4 as well).
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06e; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hardening.py tests/test_gpu_fullsize.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -4 $O/tests.log
M="--steps 20 --warmup 5 --no-cpu-baseline --no-train --no-render"
for i in 1 2 3; do
  for wl in C2 C3 C4; do
    timeout 200 python tools/bench_variant.py build_variants/libmvp_r06mask.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('mask', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new ', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
  done
done
