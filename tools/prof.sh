#!/bin/bash
# tools/prof.sh <tag> [bench args...] -- run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py,
# keeping only the small CSV summaries under gpurun_out/<tag>/ (copy the ones to keep into profiles/).
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=/tmp/prof_$TAG; rm -rf $OUT; mkdir -p $OUT gpurun_out/$TAG
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python bench.py "$@" > gpurun_out/$TAG/bench.log 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} gpurun_out/$TAG/ \;
tail -1 gpurun_out/$TAG/bench.log | cut -c1-400
python - gpurun_out/$TAG/${TAG}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
out = [rows[0]] + [[r[0][:90]] + r[1:] for r in rows[1:]]
csv.writer(open(sys.argv[1], "w")).writerows(out)
for r in out[:9]:
    print(" | ".join(x[:70] for x in r[:6]))
PY
