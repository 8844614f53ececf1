#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_bgmlp.py -m gpu -q 2>&1 | tail -2
