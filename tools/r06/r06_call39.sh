#!/bin/bash
# round-6 GPU call 39: pose sums about the ray's first sample in the box instead of the ray origin: seed 3044 replayed (and a few
# others), the whole GPU suite, A/B against the previous build.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06ad; mkdir -p $O
timeout 400 python tools/diag_fuzz_replay_pose.py 3044 1847 339 7 2>&1 | grep -v amdgpu.ids | cut -c1-500 | tee $O/pose_replay.txt
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -3 $O/tests.log
M="--steps 20 --warmup 5 --no-cpu-baseline --no-train --no-render --no-workloads"
for i in 1 2 3; do
  for wl in "C2" "C3" "C4"; do
    timeout 200 python tools/bench_variant.py build_variants/libmvp_prev.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('prev', '$wl', d['ms_per_step'], d['kernel_ms']['march_backward'])" | tee -a $O/ab.txt
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new ', '$wl', d['ms_per_step'], d['kernel_ms']['march_backward'])" | tee -a $O/ab.txt
  done
done
