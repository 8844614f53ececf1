#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-render"
cp ava-256_amd/libmvp_gfx950.so /tmp/prod.so
for v in win0 win2 win6 prod; do
  if [ $v = prod ]; then cp /tmp/prod.so ava-256_amd/libmvp_gfx950.so; else cp build_variants/libmvp_$v.so ava-256_amd/libmvp_gfx950.so; fi
  for w in "--workload C2"; do echo -n "$v $w: "; timeout 300 $B $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f' % d['ms_per_step'], {k: round(v,3) for k,v in d['kernel_ms'].items()})"; done
done
cp /tmp/prod.so ava-256_amd/libmvp_gfx950.so
