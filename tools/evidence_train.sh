#!/bin/bash
# tools/evidence_train.sh <tag> -- the train / background-MLP half of tools/evidence.sh, for a change that leaves the march
# kernels alone: the default bench line, the MFMA counter passes of the train legs (C3 and C2 + background MLP), kernel
# stats of the C3 train leg, the fused-MLP microbenchmark with its counters, and the GPU tests that run the MLP kernels.
set -u
TAG=$1
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_bgmlp.py tests/test_trainloop.py tests/test_trainstep_parity.py -m gpu -x -q > $O/tests.log 2>&1 < /dev/null; tail -2 $O/tests.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 < /dev/null | tail -1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null; echo "bench rc $?"
bash tools/pmc_all.sh ${TAG}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" --mode train --workload C3 --steps 4 --warmup 2 > $O/mfma.log 2>&1 < /dev/null
bash tools/pmc_all.sh ${TAG}_mfma_c2bg "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" --mode train --workload C2 --bg on --steps 2 --warmup 1 > $O/mfma_c2bg.log 2>&1 < /dev/null
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_train; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o train -- python bench.py --mode train --workload C3 --steps 6 --warmup 2 > $O/train_prof.log 2>&1 < /dev/null
find /tmp/prof_train -name "*kernel_stats.csv" -exec cp {} $O/train_C3_kernel_stats.csv \;
python - $O/train_C3_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
out = [rows[0]] + [[r[0][:110]] + r[1:] for r in rows[1:40]]
csv.writer(open(sys.argv[1], "w")).writerows(out)
for r in out[:8]: print(" | ".join(x[:60] for x in r[:5]))
PY
timeout 300 python tools/bench_bgmlp_fused.py 4 512 512 > $O/bgmlp_bench.json 2>/dev/null < /dev/null; cat $O/bgmlp_bench.json
cut -c1-300 $O/bench.json
