#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02x; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "warp" > $O/pytest_warp.log 2>&1; echo "pytest warp rc $?"
tail -8 $O/pytest_warp.log
timeout 300 python tools/bench_warp.py 4 512 512 4096 > $O/warp_bench.json 2> $O/warp_bench.err; cat $O/warp_bench.json; tail -3 $O/warp_bench.err
