#!/bin/bash
# tools/exp_variants.sh <n>... -- cross-compile timing-experiment variants of the library (-DMVP_EXP=n) into
# build_variants/libmvp_exp<n>.so.  On the GPU box: cp build_variants/libmvp_exp<n>.so ava-256_amd/libmvp_gfx950.so
set -eu
cd "$(dirname "$0")/.."
mkdir -p build_variants
for n in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -fno-slp-vectorize -fno-gpu-rdc -Wno-unused-function \
    -DMVP_EXP=$n -I include -I ava-256_amd/csrc ava-256_amd/csrc/{raydirs,aabb,march,assemble,placement,gradclip,bgmlp,pixeltail,primpose,abi_misc}.hip \
    -o build_variants/libmvp_exp$n.so &
done
wait
ls -la build_variants
