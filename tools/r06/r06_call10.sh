#!/bin/bash
# round-6 GPU call 10: the 400-seed randomized parity run (all three backward owners, warp fields, image-filling boxes, drawn
# upstream-gradient shapes) on the round's kernels: 16-byte list records + mask-compacted phase 1, interleaved pose sums, rounds
# cut to the noise budget.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06fuzz; mkdir -p $O
MVP_FUZZ_SEEDS=400 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomized -p no:cacheprovider > $O/fuzz400.log 2>&1 < /dev/null; echo "fuzz rc $?"; tail -3 $O/fuzz400.log
