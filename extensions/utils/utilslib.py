"""Drop-in for the reference's compiled `utilslib` (utils.cpp:134-137): same function names and positional signatures
over the gfx950 C ABI, so the reference's unmodified extensions/utils/utils.py can `from . import utilslib`."""
from ava256_amd.native_shim import compute_raydirs_backward, compute_raydirs_forward  # noqa: F401
