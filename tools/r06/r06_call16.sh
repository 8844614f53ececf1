#!/bin/bash
# round-6 GPU call 16: which primitives of a TRAINING backward go to the two-pass kernel when the background MLP is on (C3 batch:
# 4 frames; the 80-frame batch with the MLP needs 65 GB), and what the upstream gradient looks like per channel.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06q; mkdir -p $O
timeout 300 python tools/diag_two_pass_marks.py C3 bg 2>&1 < /dev/null | tail -6 | tee $O/diag_C3_bg.txt
timeout 300 python tools/diag_two_pass_marks.py C3 2>&1 < /dev/null | tail -6 | tee $O/diag_C3.txt
