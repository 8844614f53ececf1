#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
bash tools/evidence.sh r05m > gpurun_out/r05m_evidence.log 2>&1; tail -5 gpurun_out/r05m_evidence.log | cut -c1-300
MVP_FUZZ_SEEDS=400 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k randomized > gpurun_out/r05m_fuzz400.log 2>&1; tail -3 gpurun_out/r05m_fuzz400.log
cp gpurun_out/parity_masks.json gpurun_out/r05m_fuzz400_masks.json
