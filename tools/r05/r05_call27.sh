#!/bin/bash
# bg MLP timing decomposition on the deferred-store kernels: product, no epilogue (1), no MFMA (2), no k-loop barriers (3), no fragment reads (4)
O=gpurun_out/r05x6; mkdir -p $O
for r in 1 2; do
  timeout 100 python tools/bench_bgmlp_fused.py 4 512 512 ava-256_amd/libmvp_gfx950.so 2>/dev/null < /dev/null | cut -c1-230
  for e in 1 2 3 4; do timeout 100 python tools/bench_bgmlp_fused.py 4 512 512 build_variants/libmvp_bgexp$e.so 2>/dev/null < /dev/null | cut -c1-230; done
done | tee $O/decomp.txt
