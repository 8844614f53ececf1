#!/bin/bash
# bg MLP: inference instantiation without store code + straight-line (clamped-row) store slices: tests, fused time x3, cycles
O=gpurun_out/r05x10; mkdir -p $O
timeout 300 python -m pytest tests/test_bgmlp.py tests/test_trainloop.py -m gpu -x -q > $O/tests.log 2>&1 < /dev/null; tail -2 $O/tests.log
for i in 1 2 3; do timeout 200 python tools/bench_bgmlp_fused.py 4 512 512 2>$O/bench.err < /dev/null | tee $O/bgmlp_bench.json | cut -c1-200; done
bash tools/pmc_cmd.sh r05x10_a "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" bgmlp -- python tools/bench_bgmlp_fused.py 4 512 512 < /dev/null
