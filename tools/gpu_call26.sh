#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02x; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest rc $?"
tail -6 $O/pytest_all.log
