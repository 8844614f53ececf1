#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
C="python tools/bench_bgmlp_fused.py 4 512 512"
bash tools/pmc_cmd.sh r02y_p1 "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" bgmlp -- $C
bash tools/pmc_cmd.sh r02y_p2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" bgmlp -- $C
bash tools/pmc_cmd.sh r02y_p3 "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" bgmlp -- $C
bash tools/pmc_cmd.sh r02y_p4 "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" bgmlp -- $C
