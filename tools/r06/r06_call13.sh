#!/bin/bash
# round-6 GPU call 13: the saturation clip of phase 1 on a fresh debug library: smoke, the whole GPU suite, the 400-seed fuzz.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06n; mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 < /dev/null | tail -1
timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -3 $O/tests.log
MVP_FUZZ_SEEDS=400 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomized -p no:cacheprovider > $O/fuzz400.log 2>&1 < /dev/null; echo "fuzz rc $?"; tail -2 $O/fuzz400.log
