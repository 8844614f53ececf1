"""CPU-side tests: the C-ABI library builds for gfx950, loads, exports every symbol include/mvp_abi.h
declares, validates arguments before touching a device, and the Python operator surface mirrors the
reference's (names, defaults, error behaviour).  No compute call is made here."""
import ctypes
import inspect
import os
import re

import pytest
import torch

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__  # noqa: F401  (puts ROOT on sys.path)
    import importlib.util
    spec = importlib.util.spec_from_file_location("_mvp_build", os.path.join(ROOT, "ava-256_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()  # no-op when up to date; hipcc cross-compiles gfx950 without a GPU
    from ava256_amd import _lib
    return _lib.get_lib()


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "mvp_abi.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mvp_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(lib):
    from ava256_amd import _lib
    names = _declared_functions()
    assert len(names) >= 7
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names
    assert lib.mvp_abi_version() == _lib.ABI_VERSION == 17
    assert b"bad argument" in lib.mvp_error_string(-1)
    assert lib.mvp_error_string(0) == b"ok"


def test_code_object_is_gfx950_only():
    so = os.path.join(ROOT, "ava-256_amd", "libmvp_gfx950.so")
    blob = open(so, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_70", b"sm_80"):
        assert other not in blob


def test_half_render_entry_points_validate_arguments(lib):
    """mvp_march_render_half / mvp_template_to_half / mvp_template_assemble_forward_half (include/mvp_abi.h): forward only,
    8^3 slabs, one of the two ray forms, aligned pointers -- all decided before any device work."""
    buf = (ctypes.c_float * 64)()
    p16 = (ctypes.addressof(buf) + 15) & ~15
    n = None
    # N,H,W,K | raypos,raydir,tminmax | campos,camrot,focal,princpt,pixelcoords | volradius,stepsize |
    # nodeaabb,primpos,primrot,primscale | TD,TH,TW | tplate_half,rayrgba | fadescale,fadeexp | diag,stream
    a = [1, 1, 1, 1, p16, p16, p16, n, n, n, n, n, 1.0, 0.1, p16, p16, p16, p16, 8, 8, 8, p16, p16, 8.0, 8.0, n, n]
    bad = list(a); bad[21] = None
    assert lib.mvp_march_render_half(*bad) == -1            # no slab pointer
    bad = list(a); bad[21] = p16 + 8
    assert lib.mvp_march_render_half(*bad) == -1            # misaligned slabs
    bad = list(a); bad[18] = 4
    assert lib.mvp_march_render_half(*bad) == -2            # 8^3 slabs only
    bad = list(a); bad[7] = p16
    assert lib.mvp_march_render_half(*bad) == -1            # ray tensors AND cameras
    bad = list(a); bad[0] = 0
    assert lib.mvp_march_render_half(*bad) == 0             # empty batch
    assert lib.mvp_template_to_half(-2, p16, p16, n) == -1
    assert lib.mvp_template_to_half(3, p16, p16, n) == -1   # odd voxel count
    assert lib.mvp_template_to_half(0, n, n, n) == 0
    assert lib.mvp_template_to_half(2, n, p16, n) == -1
    assert lib.mvp_template_assemble_forward_half(1, 2, 8, n, n, n, n) == -1
    assert lib.mvp_template_assemble_forward_half(0, 2, 8, n, n, n, n) == 0


def test_argument_validation_happens_before_any_device_work(lib):
    null = None
    assert lib.mvp_raydirs_forward(-1, 4, 4, null, null, null, null, null, 1.0, null, null, null, null) == -1
    assert lib.mvp_raydirs_forward(0, 4, 4, null, null, null, null, null, 1.0, null, null, null, null) == 0
    assert lib.mvp_raydirs_forward(1, 4, 4, null, null, null, null, null, 1.0, null, null, null, null) == -1
    assert lib.mvp_aabb_build(1, -2, null, null, null, null, null) == -1
    assert lib.mvp_aabb_build(0, 8, null, null, null, null, null) == 0
    assert lib.mvp_aabb_build(1, 8, null, null, null, null, null) == -1
    args = [1, 8, 8, 4, null, null, 0.1, null, null, null, null, null, 8, 8, 8, null, 0, 0, 0, null, null, null, null,
            null, null, 0, 8.0, 8.0, null, null]
    assert lib.mvp_march_forward(*args) == -1              # null pointers
    args[0] = 0
    assert lib.mvp_march_forward(*args) == 0               # empty batch
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)
    p16 = (p + 15) & ~15
    a = [1, 1, 1, 1, p16, p16, 0.0, p16, p16, p16, p16, p16, 8, 8, 8, p16, 0, 0, 0, None, p16, None, None, None, None,
         0, 8.0, 8.0, None, None]
    assert lib.mvp_march_forward(*a) == -1                 # stepsize must be > 0
    a[6] = 0.1
    a[12] = 1
    assert lib.mvp_march_forward(*a) == -2                 # slab dimension < 2: unsupported
    a[12] = 8
    a[15] = p16 + 4
    assert lib.mvp_march_forward(*a) == -1                 # misaligned template
    a[15] = p16
    a[23] = p16                                            # primlist_count without primlist
    assert lib.mvp_march_forward(*a) == -1
    a[23] = None
    a[16], a[19] = 1, p16                                  # warp grid dimension < 2
    assert lib.mvp_march_forward(*a) == -2
    bw = [1, 8, 8, 4, null, null, 0.1, null, null, null, null, null, 8, 8, 8, null, 0, 0, 0, null, null, null, null,
          null, 0, null, null, null, null, null, null, 8.0, 8.0, null, null]
    assert lib.mvp_march_backward(*bw) == -1               # null pointers
    bw[3] = 0
    bw[4] = bw[5] = bw[7] = p16
    assert lib.mvp_march_backward(*bw) == 0                # K == 0: nothing to differentiate
    # fused background MLP: B, HW | samplecoords, bias1, w1pos, wh, bh, w6, b6, acts, x0, out | stream
    assert lib.mvp_bgmlp_forward(-1, 4, *([null] * 11)) == -1
    assert lib.mvp_bgmlp_forward(0, 4, *([null] * 11)) == 0                       # no pixels: nothing to do
    assert lib.mvp_bgmlp_forward(1, 4, *([null] * 11)) == -1                      # null pointers
    assert lib.mvp_bgmlp_forward(1, 4, p16, p16, p16 + 2, p16, p16, p16, p16, null, null, p16, null) == -1   # misaligned weights
    assert lib.mvp_bgmlp_backward(1, 4, *([null] * 7)) == -1
    assert lib.mvp_bgmlp_backward(0, 0, *([null] * 7)) == 0


def test_operator_surface_mirrors_the_reference():
    import ava256_amd as ops
    sig = inspect.signature(ops.mvpraymarch)
    expect = ["raypos", "raydir", "stepsize", "tminmax", "primtransf", "template", "warp", "rayterm", "algo",
              "usebvh", "sortprims", "randomorder", "maxhitboxes", "synchitboxes", "chlast", "fadescale", "fadeexp",
              "accum", "termthresh", "griddim", "blocksize", "bwdblocksize"]  # mvpraymarch.py:295-318
    assert list(sig.parameters) == expect
    d = {k: v.default for k, v in sig.parameters.items() if v.default is not inspect.Parameter.empty}
    assert d == dict(rayterm=None, algo=0, usebvh="fixedorder", sortprims=False, randomorder=False, maxhitboxes=512,
                     synchitboxes=True, chlast=True, fadescale=8.0, fadeexp=8.0, accum=0, termthresh=0.0, griddim=3,
                     blocksize=(8, 16), bwdblocksize=(8, 16))
    # Raymarcher filters renderoptions by mvpraymarch.__code__.co_varnames (mvpraymarcher.py:45)
    for n in expect:
        assert n in ops.mvpraymarch.__code__.co_varnames
    assert list(inspect.signature(ops.compute_raydirs).parameters) == ["viewpos", "viewrot", "focal", "princpt",
                                                                      "pixelcoords", "volradius"]
    rm = ops.Raymarcher(256.0)
    assert rm.volume_radius == 256.0 and rm.dt == 1.0 / 256.0
    assert ops.Raymarcher(256.0, dt=2.0).dt == 2.0 / 256.0
    assert len(rm.state_dict()) == 0 and len(list(rm.parameters())) == 0   # checkpoints stay loadable
    assert list(inspect.signature(rm.forward).parameters) == ["raypos", "raydir", "tminmax", "decout",
                                                              "renderoptions", "rayterm", "with_pos_img"]


def test_reference_import_paths_resolve_to_this_build():
    from extensions.mvpraymarch.mvpraymarch import mvpraymarch as a
    from extensions.utils.utils import compute_raydirs as b
    from models.raymarchers.mvpraymarcher import Raymarcher as c
    import ava256_amd as ops
    assert a is ops.mvpraymarch and b is ops.compute_raydirs and c is ops.Raymarcher


def test_no_cpu_fallback():
    """CPU tensors are refused (the reference raises through AT_ASSERTM, mvpraymarch.cpp:102-104)."""
    import ava256_amd as ops
    N, H, W, K = 1, 4, 4, 2
    z = lambda *s: torch.zeros(*s)
    with pytest.raises(RuntimeError):
        ops.compute_raydirs(z(N, 3), z(N, 3, 3), z(N, 2), z(N, 2), z(N, H, W, 2), 256.0)
    with pytest.raises(RuntimeError):
        ops.mvpraymarch(z(N, H, W, 3), z(N, H, W, 3), 0.1, z(N, H, W, 2), (z(N, K, 3), z(N, K, 3, 3), z(N, K, 3)),
                        z(N, K, 8, 8, 8, 4), None)
    # and nothing in the product package imports the oracle
    pkg = os.path.join(ROOT, "ava-256_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            assert "oracle" not in open(os.path.join(pkg, fn)).read().replace("no oracle", ""), fn


def test_list_capacity_policy_on_the_host():
    """Host logic of the forward->backward list capacity (ava-256_amd/mvpraymarch.py): the first call of a shape uses the
    heuristic, later calls 1.25 x the measured demand, always a multiple of 8 (the library reads lists 32 bytes at a time
    and rejects capacities that are not a multiple of 4), at least 32, at most 2048.  The demand is robust: one outlier
    primitive does not size everybody's list, a spike decays, the capacity does not flutter, and N*K*cap*16 bytes stay
    inside a budget."""
    import importlib
    op = importlib.import_module("ava256_amd.mvpraymarch")
    dev = torch.device("cuda", 0)   # only its index is used as a key
    key = (0, 512, 512, 4096)
    op._LIST_DEMAND.pop(key, None)
    cap0 = op.primlist_capacity(512, 512, 4096, dev)
    assert cap0 == 40 and cap0 % 8 == 0                        # 4 x 10 x 4096 packets / 4096 primitives
    assert op.primlist_capacity(64, 64, 16384, dev) == 32      # never below 32
    assert op.primlist_capacity(4096, 4096, 16, dev) == 2048   # never above 2048
    for demand, want in ((10, 32), (61, 80), (100, 128), (5000, 2048)):
        st = op._LIST_DEMAND[key] = op._ListDemand()
        st.note(demand)
        cap = op.primlist_capacity(512, 512, 4096, dev)
        assert cap == want and cap % 8 == 0 and cap >= min(demand, 2048), (demand, cap)
    op._LIST_DEMAND.pop(key, None)
    assert op.primlist_capacity(512, 512, 4096, None) == cap0  # no device: the heuristic

    def hist_of(counts):
        h = [0] * 257
        for c in counts:
            h[min(c, 2047) >> 3] += 1
        h[256] = max(counts)
        return h

    # the whole distribution moved (close-up camera): the maximum is what is wanted
    assert op.wanted_from_histogram(hist_of([60] * 3000 + [90] * 1000 + [101])) == 101
    # ONE image-filling primitive among 4096 (the advisor's case): sized for 2 x the 99.9th percentile, not for it
    w = op.wanted_from_histogram(hist_of([12] * 4000 + [20] * 95 + [1900]))
    assert 32 <= w <= 64, w
    assert op.wanted_from_histogram([0] * 257) == 0
    # a spike is followed at once and forgotten at 10 % per measurement; the capacity in use moves only out of [0.6, 1] x
    st = op._LIST_DEMAND[key] = op._ListDemand()
    st.note(20)
    c_norm = op.primlist_capacity(512, 512, 4096, dev)
    st.note(1500)
    c_spike = op.primlist_capacity(512, 512, 4096, dev)
    assert c_norm == 32 and c_spike == 1880
    seen = []
    for _ in range(60):
        st.note(20)
        seen.append(op.primlist_capacity(512, 512, 4096, dev))
    assert seen[0] == c_spike and seen[-1] <= 48 and sorted(seen, reverse=True) == seen
    assert len(set(seen)) <= 12, sorted(set(seen))              # steps, not a new allocation size per call
    # memory budget (16-byte entries since round 6): at C2 (80 x 4096 primitives) the lists never pass
    # max(128 MiB, 4 KiB per primitive) = 1280 MiB
    st.note(1500)
    capb = op.primlist_capacity(512, 512, 4096, dev, N=80)
    assert op.LIST_ENTRY_WORDS == 4 and capb == 256 and 80 * 4096 * capb * 16 <= 4096 * 80 * 4096
    assert op.primlist_capacity(512, 512, 4096, dev, N=1) == 1880   # one image: 128 MiB allow it
    op._LIST_DEMAND.pop(key, None)


def test_half_slab_operators_have_no_cpu_path():
    """ava-256_amd/halfslab.py: like every operator of this build, the opt-in fp16 render path refuses CPU tensors, wrong dtypes
    and inputs that require gradients -- before it touches the library."""
    from ava256_amd import halfslab
    t = torch.zeros(1, 2, 8, 8, 8, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        halfslab.template_to_half(t)
    with pytest.raises(RuntimeError, match="no CPU path"):
        halfslab.assemble_template_half(torch.zeros(1, 24, 16, 16), torch.zeros(1, 8, 16, 16), 4)
    prim = (torch.zeros(1, 2, 3), torch.zeros(1, 2, 3, 3), torch.ones(1, 2, 3))
    with pytest.raises(RuntimeError, match="no CPU path"):
        halfslab.render_half(torch.zeros(1, 4, 4, 3), torch.zeros(1, 4, 4, 3), 0.1, torch.zeros(1, 4, 4, 2), prim,
                             t.to(torch.float16))
    with pytest.raises(RuntimeError, match="no CPU path|renders only"):
        halfslab.render_half_from_cameras(torch.zeros(1, 3, requires_grad=True), torch.zeros(1, 3, 3), torch.ones(1, 2),
                                          torch.zeros(1, 2), (4, 4), 1.0, 0.1, prim, t.to(torch.float16))
