#!/bin/bash
# what the bg-MLP kernels do outside their MFMAs: three counter passes on tools/bench_bgmlp_fused.py 4 512 512
O=gpurun_out/r05y; mkdir -p $O
CMD="python tools/bench_bgmlp_fused.py 4 512 512"
bash tools/pmc_cmd.sh r05y_a "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" bgmlp -- $CMD > $O/a.txt 2>&1 < /dev/null
bash tools/pmc_cmd.sh r05y_b "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" bgmlp -- $CMD > $O/b.txt 2>&1 < /dev/null
bash tools/pmc_cmd.sh r05y_c "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_DATA_FIFO_FULL" bgmlp -- $CMD > $O/c.txt 2>&1 < /dev/null
cat $O/a.txt $O/b.txt $O/c.txt
