#!/bin/bash
set -u
TAG=r02z
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$TAG; mkdir -p $O
bash tools/prof.sh $TAG --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-render > $O/prof.log 2>&1; tail -4 $O/prof.log
M="--steps 3 --warmup 1 --no-cpu-baseline --no-train --no-render"
bash tools/pmc.sh ${TAG}_fetch "FETCH_SIZE" $M > $O/fetch.log 2>&1
bash tools/pmc.sh ${TAG}_write "WRITE_SIZE" $M > $O/write.log 2>&1
bash tools/pmc.sh ${TAG}_sq "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" $M > $O/sq.log 2>&1
bash tools/pmc.sh ${TAG}_ta "TA_TA_BUSY_sum TA_BUSY_max TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" $M > $O/ta.log 2>&1
bash tools/pmc.sh ${TAG}_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" $M > $O/lds.log 2>&1
grep -h "march_kernel<false\|bwd_prim" gpurun_out/${TAG}_fetch/pmc_summary.csv gpurun_out/${TAG}_write/pmc_summary.csv
