#!/bin/bash
# round-6 GPU call 1: (a) the GPU suite on the forward that appends its list entries AFTER the sweep; (b) A/B of that forward
# against round 5's library (build_variants/libmvp_r05base.so) at C2 / C3 / C4, three interleaved rounds; (c) the counter
# passes that say what binds the SHIPPED bwd_prim_kernel incl. lane utilisation; (d) FETCH / WRITE passes
# for C3, C4 and the saturated scene (item 4a).
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06a; mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 < /dev/null | tail -1
timeout 400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -4 $O/tests.log
M="--steps 20 --warmup 5 --no-cpu-baseline --no-train --no-render"
for i in 1 2 3; do
  for wl in C2 C3 C4; do
    timeout 200 python tools/bench_variant.py build_variants/libmvp_r05base.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('base', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('new ', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/ab.txt
  done
done
P="--steps 3 --warmup 1 --no-cpu-baseline --no-train --no-render"
bash tools/pmc.sh r06a_bind1 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_WAVES SQ_ACTIVE_INST_VALU" $P > $O/bind1.log 2>&1; grep bwd_prim_kernel $O/bind1.log
bash tools/pmc.sh r06a_bind2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_LEVEL_WAVES GRBM_GUI_ACTIVE" $P > $O/bind2.log 2>&1; grep bwd_prim_kernel $O/bind2.log
bash tools/pmc.sh r06a_bind3 "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SALU" $P > $O/bind3.log 2>&1; grep bwd_prim_kernel $O/bind3.log
bash tools/pmc.sh r06a_bind4 "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" $P > $O/bind4.log 2>&1; grep bwd_prim_kernel $O/bind4.log
bash tools/pmc.sh r06a_bind5 "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VALU GRBM_GUI_ACTIVE" $P > $O/bind5.log 2>&1; grep "mvp" $O/bind5.log
for wl in C3 C4; do
  bash tools/pmc.sh r06a_fetch_$wl "FETCH_SIZE" $P --workload $wl > $O/fetch_$wl.log 2>&1; grep mvp $O/fetch_$wl.log
  bash tools/pmc.sh r06a_write_$wl "WRITE_SIZE" $P --workload $wl > $O/write_$wl.log 2>&1; grep mvp $O/write_$wl.log
done
bash tools/pmc.sh r06a_fetch_sat "FETCH_SIZE" $P --alpha-gain 40 > $O/fetch_sat.log 2>&1; grep mvp $O/fetch_sat.log
bash tools/pmc.sh r06a_write_sat "WRITE_SIZE" $P --alpha-gain 40 > $O/write_sat.log 2>&1; grep mvp $O/write_sat.log
bash tools/pmc.sh r06a_fetch "FETCH_SIZE" $P > $O/fetch.log 2>&1; grep mvp $O/fetch.log
bash tools/pmc.sh r06a_write "WRITE_SIZE" $P > $O/write.log 2>&1; grep mvp $O/write.log
