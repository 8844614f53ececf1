#!/bin/bash
# round-6 GPU call 27: the whole GPU suite and the 400-seed fuzz (one draw in four has a warp field) on the branch-free
# warp-field kernels.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06x; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -5 $O/tests.log
MVP_FUZZ_SEEDS=400 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomized -p no:cacheprovider > $O/fuzz400.log 2>&1 < /dev/null; echo "fuzz rc $?"; tail -8 $O/fuzz400.log
