#!/usr/bin/env python3
"""tools/pytest_variant.py <library.so> [pytest arguments...] -- run (part of) the GPU test suite against a NON-product build
of the same ABI (build_variants/libmvp_*.so: a parked experiment rebuilt by tools/build_patched.sh), so that a candidate
kernel is held to the parity tests before it is measured.  The product path never does this."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    lib = os.path.abspath(sys.argv[1])
    from ava256_amd import _lib
    _lib.use_library(lib)
    _orig = _lib.use_library

    def keep_variant(path=None):  # tests that switch to the debug build go back to the VARIANT, not to the product
        _orig(path or lib)
    _lib.use_library = keep_variant
    import pytest
    print("# library under test:", lib, file=sys.stderr)
    raise SystemExit(pytest.main(sys.argv[2:]))
