#!/bin/bash
# round-2 evidence run: kernel stats of the default bench line (march + train legs), HBM traffic passes, MFMA counters of the
# train leg (C3: 4 frames, bf16 background MLP)
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02h; mkdir -p $O
bash tools/prof.sh r02h --steps 5 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1; tail -12 $O/prof.log
bash tools/pmc.sh r02h_fetch "FETCH_SIZE" --steps 3 --warmup 1 --no-cpu-baseline --no-train > $O/fetch.log 2>&1; cat $O/fetch.log | tail -8
bash tools/pmc.sh r02h_write "WRITE_SIZE" --steps 3 --warmup 1 --no-cpu-baseline --no-train > $O/write.log 2>&1; cat $O/write.log | tail -8
bash tools/pmc_all.sh r02h_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" --mode train --workload C3 --steps 4 --warmup 2 > $O/mfma.log 2>&1; tail -14 $O/mfma.log
timeout 300 python bench.py --mode train --workload C3 --steps 10 --warmup 3 > $O/train_C3.json 2>$O/train_C3.err; cut -c1-400 $O/train_C3.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_train; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o train -- python bench.py --mode train --workload C3 --steps 6 --warmup 2 > $O/train_prof.log 2>&1
find /tmp/prof_train -name "*kernel_stats.csv" -exec cp {} $O/train_C3_kernel_stats.csv \;
head -14 $O/train_C3_kernel_stats.csv | cut -c1-160
