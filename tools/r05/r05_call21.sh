#!/bin/bash
# bg MLP with two 96-pixel 4-wave workgroups per CU: parity tests, fused fwd / fwd+bwd time, per-kernel time
O=gpurun_out/r05x; mkdir -p $O
timeout 300 python -m pytest tests/test_bgmlp.py -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 200 python tools/bench_bgmlp_fused.py 4 512 512 > $O/bgmlp_bench.json 2>$O/bench.err; cat $O/bgmlp_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bg -- python $GRAFT_REPO_ROOT/tools/bench_bgmlp_fused.py 4 512 512 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $O/prof -name "*kernel_stats.csv" | head -1); grep -i "bgmlp" $f | cut -c1-120
