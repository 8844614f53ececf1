#!/bin/bash
# tools/prof.sh <tag> [bench args...] -- run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py,
# keeping only the small CSV summaries under gpurun_out/<tag>/ (copy the ones to keep into profiles/).
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=/tmp/prof_$TAG; rm -rf $OUT; mkdir -p $OUT gpurun_out/$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o $TAG -- python bench.py "$@" > gpurun_out/$TAG/bench.log 2>&1
find $OUT -name "*stats*.csv" -exec cp {} gpurun_out/$TAG/ \;
find $OUT -type f | head -20 > gpurun_out/$TAG/files.txt
tail -1 gpurun_out/$TAG/bench.log | cut -c1-600
for f in gpurun_out/$TAG/*kernel_stats.csv; do echo "== $f"; head -12 "$f"; done
