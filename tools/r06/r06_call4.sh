#!/bin/bash
# round-6 GPU call 4: pose sums of the backward as ONE interleaved 12-way DPP scan (72 VALU instead of ~253 per wave):
# backward parity tests, then A/B against the mask-compaction build (build_variants/libmvp_r06mask.so) and round 5's library.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06d; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hardening.py tests/test_gpu_fullsize.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -4 $O/tests.log
M="--steps 20 --warmup 5 --no-cpu-baseline --no-train --no-render"
for i in 1 2 3; do
  for wl in C2 C3 C4; do
    MVP_VARIANT_ABI=15 timeout 200 python tools/bench_variant.py build_variants/libmvp_r05base.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('r05 ', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
    timeout 200 python tools/bench_variant.py build_variants/libmvp_r06mask.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('mask', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new ', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
  done
done
