#!/bin/bash
# round-5 GPU call 1: ubenches (packed 64-bit LDS atomics, fp16 gather layouts), the counters that say what binds
# bwd_prim_kernel, the list of counters this box offers, and a baseline bench line of HEAD.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r05a; mkdir -p $O
./tools/ubench/lds_pack64 > $O/ubench_lds_pack64.txt 2>&1; cat $O/ubench_lds_pack64.txt
./tools/ubench/gather_layout > $O/ubench_gather_layout.txt 2>&1; cat $O/ubench_gather_layout.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
(rocprofv3 --list-avail 2>&1 || rocprofv3 -L 2>&1) | grep -o "\bSQ_[A-Z0-9_]*\|\bTA_[A-Z0-9_]*\|\bTCP_[A-Z0-9_]*\|\bTCC_[A-Z0-9_]*\|\bGRBM_[A-Z0-9_]*" | sort -u > $O/counters_avail.txt; wc -l $O/counters_avail.txt
M="--steps 3 --warmup 1 --no-cpu-baseline --no-train --no-render"
bash tools/pmc.sh r05a_bind1 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_WAVES SQ_ACTIVE_INST_VALU" $M > $O/bind1.log 2>&1; grep bwd_prim_kernel $O/bind1.log
bash tools/pmc.sh r05a_bind2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LEVEL_WAVES GRBM_GUI_ACTIVE" $M > $O/bind2.log 2>&1; grep bwd_prim_kernel $O/bind2.log
bash tools/pmc.sh r05a_bind3 "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SALU" $M > $O/bind3.log 2>&1; grep bwd_prim_kernel $O/bind3.log
bash tools/pmc.sh r05a_bind4 "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" $M > $O/bind4.log 2>&1; grep bwd_prim_kernel $O/bind4.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-400 $O/bench.json
