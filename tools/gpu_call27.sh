#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02y; mkdir -p $O
timeout 600 python -m pytest tests/test_bgmlp.py -m gpu -q > $O/pytest_bgmlp.log 2>&1; echo "pytest rc $?"
grep -n "Error\|assert\|passed\|failed" $O/pytest_bgmlp.log | head -30
timeout 300 python tools/bench_bgmlp_fused.py 4 512 512 > $O/bgmlp_bench.json 2> $O/bgmlp_bench.err; cat $O/bgmlp_bench.json; tail -3 $O/bgmlp_bench.err
