#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r02y; mkdir -p $O
cp ava-256_amd/libmvp_gfx950.so /tmp/prod.so
for v in prod bg1 bg2 bg5 bg6; do
  if [ $v = prod ]; then cp /tmp/prod.so ava-256_amd/libmvp_gfx950.so; else cp build_variants/libmvp_$v.so ava-256_amd/libmvp_gfx950.so; fi
  echo -n "$v: "; timeout 120 python tools/bench_bgmlp_fused.py 4 512 512 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['fused'])"
done
cp /tmp/prod.so ava-256_amd/libmvp_gfx950.so
