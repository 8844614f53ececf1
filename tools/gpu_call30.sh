#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02y; mkdir -p $O
rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python tools/bench_bgmlp_fused.py 4 512 512 > $O/kt.log 2>&1
F=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); cp $F $O/bgmlp_kernel_stats.csv
python - $F <<'PY'
import sys
for i, l in enumerate(open(sys.argv[1])):
    if i > 22: break
    p = l.rstrip().rsplit(',', 7)
    print(p[0][:70].ljust(70), p[1:5])
PY
