#!/bin/bash
# round-6 GPU call 34: 1500 further fuzz draws (seeds 1000..2499) on the final kernels.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06x; mkdir -p $O
MVP_FUZZ_FIRST=1000 MVP_FUZZ_SEEDS=1500 timeout 3000 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomized -p no:cacheprovider > $O/fuzz_1000_2499.log 2>&1 < /dev/null; echo "fuzz rc $?"; tail -25 $O/fuzz_1000_2499.log | cut -c1-300
