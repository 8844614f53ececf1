#!/bin/bash
# round-6 GPU call 25: the warp-field forward through the lane-independent sweep with the branch-free sampler (8 dwordx3 + 8
# dwordx4 gathers per sample, one wait each): warp tests, fuzz draws, tools/bench_warp.py (plain / prim / ray).
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06w; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "warp or fuzz or random or golden or hardening" > $O/tests_warp.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -5 $O/tests_warp.log
for i in 1 2; do timeout 300 python tools/bench_warp.py 4 512 512 4096 2>/dev/null < /dev/null | tee -a $O/warp_bench.txt; done
timeout 300 python tools/bench_warp.py 4 512 512 16384 2>/dev/null < /dev/null | tee -a $O/warp_bench.txt
