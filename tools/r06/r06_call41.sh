#!/bin/bash
# round-6 GPU call 41: 1000 further fuzz draws (seeds 3500..4499) on the final kernels (per-ray root test, v_rcp bounds).
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06y3; mkdir -p $O
MVP_FUZZ_FIRST=3500 MVP_FUZZ_SEEDS=1000 timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomized -p no:cacheprovider > $O/fuzz_3500_4499.log 2>&1 < /dev/null; echo "fuzz rc $?"; tail -25 $O/fuzz_3500_4499.log | cut -c1-300
