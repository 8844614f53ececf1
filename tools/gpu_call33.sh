#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp ava-256_amd/libmvp_gfx950.so /tmp/prod.so
for v in prod q6 q8; do
  if [ $v = prod ]; then cp /tmp/prod.so ava-256_amd/libmvp_gfx950.so; else cp build_variants/libmvp_$v.so ava-256_amd/libmvp_gfx950.so; fi
  rm -rf /tmp/kt; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python tools/bench_bgmlp_fused.py 4 512 512 > /dev/null 2>&1
  F=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
  echo -n "$v: "; grep bgmlp $F | python -c "
import sys
for l in sys.stdin:
    p=l.rstrip().rsplit(',',7); print(p[0][6:30], 'avg %.3f ms min %.3f' % (float(p[3])/1e6, float(p[5])/1e6), end=' | ')
print()"
done
cp /tmp/prod.so ava-256_amd/libmvp_gfx950.so
