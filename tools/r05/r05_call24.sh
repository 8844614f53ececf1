#!/bin/bash
# bg MLP with the planes stored from inside the GEMM that reads them: parity tests, fused time, per-kernel cycles
O=gpurun_out/r05x3; mkdir -p $O
timeout 300 python -m pytest tests/test_bgmlp.py -m gpu -x -q > $O/tests.log 2>&1 < /dev/null; tail -3 $O/tests.log
timeout 200 python tools/bench_bgmlp_fused.py 4 512 512 > $O/bgmlp_bench.json 2>$O/bench.err < /dev/null; cat $O/bgmlp_bench.json
bash tools/pmc_cmd.sh r05x3_a "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" bgmlp -- python tools/bench_bgmlp_fused.py 4 512 512 < /dev/null
