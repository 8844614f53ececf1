"""CPU test of the march kernels' block -> work mapping (csrc/march_common.h: packet_of_block / prim_of_block), evaluated on
the host through mvp_march_block_map -- the same functions the kernels call.  For many (N, H, W, K): every (image, packet)
and every (image, primitive) is produced by exactly one block; whole images sit on one XCD (block % 8); shared images
use all the XCDs their split says."""
import ctypes
import itertools

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__  # noqa: F401
    from ava256_amd import _lib
    return _lib.get_lib()


def _map(lib, N, H, W, K, kind):
    total = ctypes.c_int(0)
    assert lib.mvp_march_block_map(N, H, W, K, kind, 0, 0, None, ctypes.byref(total)) == 0
    out = np.empty((total.value, 2), dtype=np.int32)
    if total.value:
        assert lib.mvp_march_block_map(N, H, W, K, kind, 0, total.value, out.ctypes.data, None) == 0
    return out


SHAPES = [(1, 8, 8, 1), (1, 45, 52, 7), (2, 64, 64, 300), (3, 40, 40, 129), (4, 128, 96, 512), (5, 17, 200, 130),
          (7, 33, 31, 64), (8, 64, 64, 256), (9, 24, 24, 100), (10, 44, 52, 150), (12, 16, 16, 1000), (16, 72, 40, 128),
          (17, 44, 52, 150), (23, 9, 9, 3), (80, 64, 64, 64)]


@pytest.mark.parametrize("N,H,W,K", SHAPES)
def test_every_packet_and_primitive_is_owned_by_exactly_one_block(lib, N, H, W, K):
    tx, ty = (W + 7) // 8, (H + 7) // 8
    for kind, units in ((0, tx * ty), (1, K)):
        m = _map(lib, N, H, W, K, kind)
        live = m[m[:, 0] >= 0]
        assert live[:, 0].max() < N and live[:, 1].max() < units and live.min() >= 0
        key = live[:, 0].astype(np.int64) * units + live[:, 1]
        assert len(key) == N * units and len(np.unique(key)) == N * units, (kind, len(key), N * units)
        # XCD placement: a whole image (n < N - N % 8) lives on XCD n % 8; a shared one on the XCDs its split says
        blocks = np.nonzero(m[:, 0] >= 0)[0]
        xcd = blocks % 8
        whole = N - N % 8
        for n in range(N):
            xs = np.unique(xcd[live[:, 0] == n])
            if n < whole:
                assert list(xs) == [n % 8], (n, xs)
            else:
                R = N - whole
                F = 2 if R == 4 else 4 if R == 2 else 8
                assert len(xs) <= F and (len(xs) == F or units < 3 * F * 128), (n, xs, F)


def test_argument_checks(lib):
    out = (ctypes.c_int * 4)()
    assert lib.mvp_march_block_map(-1, 8, 8, 1, 0, 0, 0, None, None) == -1
    assert lib.mvp_march_block_map(1, 8, 8, 1, 2, 0, 0, None, None) == -1
    assert lib.mvp_march_block_map(1, 8, 8, 1, 0, 0, 2, None, None) == -1
    assert lib.mvp_march_block_map(1, 8, 8, 1, 0, 10 ** 6, 2, ctypes.addressof(out), None) == 0
    assert list(out) == [-1, -1, -1, -1]                      # blocks past the grid do nothing
