#!/bin/bash
# bg MLP sign planes, straight-line slices: product tree (scheduler fences) vs variant without them; tests, fused time, cycles per kernel
O=gpurun_out/r05x9; mkdir -p $O
timeout 300 python -m pytest tests/test_bgmlp.py -m gpu -x -q > $O/tests.log 2>&1 < /dev/null; tail -2 $O/tests.log
for i in 1 2; do
timeout 200 python tools/bench_bgmlp_fused.py 4 512 512 ava-256_amd/libmvp_gfx950.so 2>$O/bench.err < /dev/null | cut -c1-230
timeout 200 python tools/bench_bgmlp_fused.py 4 512 512 build_variants/libmvp_nosb.so 2>$O/bench.err < /dev/null | cut -c1-230
done
bash tools/pmc_cmd.sh r05x9_a "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" bgmlp -- python tools/bench_bgmlp_fused.py 4 512 512 < /dev/null
bash tools/pmc_cmd.sh r05x9_b "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" bgmlp -- python tools/bench_bgmlp_fused.py 4 512 512 build_variants/libmvp_nosb.so < /dev/null
