#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02y; mkdir -p $O
MVP_FUZZ_SEEDS=200 timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "randomized" > $O/fuzz200.log 2>&1; echo "rc $?"; tail -5 $O/fuzz200.log
