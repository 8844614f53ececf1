#!/bin/bash
# tools/exp_variants.sh <n>... -- timing-experiment variants of the primitive-centric backward (-DMVP_EXP=n: 1 conflict-free
# scatter addresses, 2 no scatter atomics, 3 no march at all, 4 LDS scatter census, 6 fp32 LDS atomics, 7 integer atomics on
# raw bits -- all but 4 compute WRONG gradients, time only).  The knobs are not in the product sources: they come from
# profiles/r05_timing_variants.patch, applied to a copy (tools/build_patched.sh) -> build_variants/libmvp_exp<n>.so.
# On the GPU box: python tools/bench_variant.py build_variants/libmvp_exp<n>.so --steps 5
set -eu
cd "$(dirname "$0")/.."
for n in "$@"; do
  bash tools/build_patched.sh exp$n profiles/r05_timing_variants.patch -DMVP_EXP=$n &
done
wait
