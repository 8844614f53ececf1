#!/bin/bash
# round-6 GPU call 3: (a) the rest of the GPU suite on the mask-compacted phase 1 (call 2 stopped at a stale debug library);
# (b) instruction / time census of the SHIPPED bwd_prim_kernel by phase: census builds of profiles/r06_bwd_stage_census.patch
# (1 skeleton only, 2 + phase 1 and the queue, 3 + the walk without samples; WRONG gradients) against the product, HIP-event time
# and SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_THREAD_CYCLES_VALU / SQ_WAIT_ANY per launch, at C2 (and times at C3 / C4).
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06c; mkdir -p $O
timeout 500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1 < /dev/null; echo "pytest rc $?"; tail -4 $O/tests.log
M="--steps 10 --warmup 3 --no-cpu-baseline --no-train --no-render"
for i in 1 2; do
  for v in bstage1 bstage2 bstage3; do
    for wl in C2 C3 C4; do
      timeout 200 python tools/bench_variant.py build_variants/libmvp_$v.so $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/census_times.txt
    done
  done
  for wl in C2 C3 C4; do
    timeout 200 python bench.py $M --workload $wl 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('product', '$wl', d['ms_per_step'], d['kernel_ms'])" | tee -a $O/census_times.txt
  done
done
P="--steps 3 --warmup 1 --no-cpu-baseline --no-train --no-render"
C="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"
for v in bstage1 bstage2 bstage3; do
  bash tools/pmc_cmd.sh r06c_$v "$C" bwd_prim_kernel -- python tools/bench_variant.py build_variants/libmvp_$v.so $P | tee -a $O/census_counters.txt
done
bash tools/pmc_cmd.sh r06c_product "$C" bwd_prim_kernel -- python bench.py $P | tee -a $O/census_counters.txt
