#!/bin/bash
# round-6 GPU call 28: 600 more fuzz draws (seeds 400..999; one in four has a warp field) on the branch-free warp-field kernels.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06x; mkdir -p $O
MVP_FUZZ_FIRST=400 MVP_FUZZ_SEEDS=600 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomized -p no:cacheprovider > $O/fuzz_400_999.log 2>&1 < /dev/null; echo "fuzz rc $?"; tail -12 $O/fuzz_400_999.log
