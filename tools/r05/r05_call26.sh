#!/bin/bash
# bg MLP (no spills; bias through LDS; next tile's pixel + weights requested ahead): parity tests, fused time x2, cycles
O=gpurun_out/r05x5; mkdir -p $O
timeout 300 python -m pytest tests/test_bgmlp.py -m gpu -x -q > $O/tests.log 2>&1 < /dev/null; tail -3 $O/tests.log
for i in 1 2; do timeout 200 python tools/bench_bgmlp_fused.py 4 512 512 2>$O/bench.err < /dev/null | tee $O/bgmlp_bench.json | cut -c1-200; done
bash tools/pmc_cmd.sh r05x5_a "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" bgmlp -- python tools/bench_bgmlp_fused.py 4 512 512 < /dev/null
