#!/bin/bash
# round-6 GPU call 26: the warp-field variant of the primitive-centric backward rewritten branch-free on register pairs with
# packed 64-bit accumulators: warp / fuzz / golden / hardening tests, tools/bench_warp.py.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06x; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "warp or fuzz or random or golden or hardening" > $O/tests_warp.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -30 $O/tests_warp.log
for a in "4 512 512 4096" "4 512 512 4096" "4 512 512 16384"; do timeout 300 python tools/bench_warp.py $a 2>&1 | tail -1 | tee -a $O/warp_bench.txt; done
