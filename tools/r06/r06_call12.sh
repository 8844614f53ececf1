#!/bin/bash
# round-6 GPU call 12: phase 1 clips a ray's step range at its saturation key (steps behind the saturating sample are not queued):
# backward parity tests, then A/B against the previous build on the headline scene, the saturated scene (opacity x 40), C3, C4.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06l; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hardening.py tests/test_gpu_fullsize.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -3 $O/tests.log
M="--steps 20 --warmup 5 --no-cpu-baseline --no-train --no-render"
for i in 1 2 3; do
  for wl in "C2" "C2 --alpha-gain 40" "C2 --alpha-gain 200" "C3" "C4"; do
    timeout 200 python tools/bench_variant.py build_variants/libmvp_r06m.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('prev', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new ', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
  done
done
