#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02x; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "warp" > $O/pytest_warp2.log 2>&1; echo "pytest warp rc $?"
grep -n "AssertionError: (\|FAILED\|passed\|failed" $O/pytest_warp2.log | head -20
