#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02c; mkdir -p $O
timeout 300 python tests/debug/debug_sweeps.py > $O/debug_sweeps.log 2>&1; cat $O/debug_sweeps.log | head -60
timeout 600 python -m pytest tests/test_gpu_hardening.py tests/test_gpu_parity.py -m gpu -q -s > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -15 $O/pytest.log
# PMC passes of the new forward (product library)
bash tools/pmc.sh r02c_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" --steps 3 --warmup 1 --no-cpu-baseline
bash tools/pmc.sh r02c_ta "TA_TA_BUSY_sum TA_BUSY_max TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" --steps 3 --warmup 1 --no-cpu-baseline
bash tools/pmc.sh r02c_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" --steps 3 --warmup 1 --no-cpu-baseline
