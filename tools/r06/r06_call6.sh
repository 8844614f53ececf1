#!/bin/bash
# round-6 GPU call 6: the training scene's two-pass primitives (how many, how long their lists) and the kernel stats of the C2 train
# legs (background off / on) -- the "0.33-0.8 ms serial tail" of VERDICT r5 item 1c, before anything is changed.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06f; mkdir -p $O
timeout 300 python tools/diag_two_pass_marks.py C2 2>&1 < /dev/null | tail -8 | tee $O/diag_C2.txt
timeout 300 python tools/diag_two_pass_marks.py C3 2>&1 < /dev/null | tail -8 | tee $O/diag_C3.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for bg in off on; do
  rm -rf /tmp/prof_t; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t -o t -- python bench.py --mode train --workload C2 --bg $bg --steps 6 --warmup 3 > $O/train_C2_$bg.log 2>&1 < /dev/null
  find /tmp/prof_t -name "*kernel_stats.csv" -exec cp {} $O/train_C2_${bg}_kernel_stats.csv \;
  python - $O/train_C2_${bg}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
out = [rows[0]] + [[r[0][:100]] + r[1:] for r in rows[1:30]]
csv.writer(open(sys.argv[1], "w")).writerows(out)
for r in out[:14]: print(" | ".join(x[:64] for x in r[:6]))
PY
  tail -1 $O/train_C2_$bg.log | cut -c1-300
done
