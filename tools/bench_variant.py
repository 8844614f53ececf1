#!/usr/bin/env python3
"""tools/bench_variant.py <library.so> [bench.py arguments...] -- bench.py's march leg against a NON-product build of the
same ABI (build_variants/libmvp_*.so: timing experiments, -D knobs).  The product path never does this; the line it
prints says which library produced it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    lib = os.path.abspath(sys.argv[1])
    from ava256_amd import _lib
    if os.environ.get("MVP_VARIANT_ABI"):
        # an OLDER library of a compatible call surface (round 6: ABI 15 reads 8-byte list records out of the 16-byte ones
        # the operators allocate -- the buffer is merely larger than it needs): timing A/B only
        _lib.ABI_VERSION = int(os.environ["MVP_VARIANT_ABI"])
    _lib.use_library(lib)
    import bench
    print("# library:", lib, file=sys.stderr)
    bench.main(sys.argv[2:] + ["--no-train", "--no-cpu-baseline"])
