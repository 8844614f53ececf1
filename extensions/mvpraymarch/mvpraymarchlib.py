"""Drop-in for the reference's compiled `mvpraymarchlib` (mvpraymarch.cpp:398-405): same function names and positional
signatures over the gfx950 C ABI, so the reference's unmodified extensions/mvpraymarch/mvpraymarch.py can
`from . import mvpraymarchlib`."""
from ava256_amd.native_shim import build_tree, compute_aabb, compute_morton, raymarch_backward, raymarch_forward  # noqa: F401
