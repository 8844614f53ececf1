#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05z_smoke.log 2>&1; tail -1 gpurun_out/r05z_smoke.log
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r05z_pytest.log 2>&1; tail -2 gpurun_out/r05z_pytest.log
cp gpurun_out/parity_masks.json gpurun_out/r05z_parity_masks.json
bash tools/evidence.sh r05z > gpurun_out/r05z_evidence.log 2>&1; tail -3 gpurun_out/r05z_evidence.log | cut -c1-300
