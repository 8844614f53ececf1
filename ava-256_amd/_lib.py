"""ctypes binding of the C ABI declared in include/mvp_abi.h (libmvp_gfx950.so).

No fallback of any kind: if the library is missing the first operator call raises, and every
non-zero status returned by the library is turned into a RuntimeError.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmvp_gfx950.so")

_c_int, _c_float, _c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p

# name -> (restype, argtypes); one entry per declaration in include/mvp_abi.h
SIGNATURES = {
    "mvp_abi_version": (_c_int, []),
    "mvp_error_string": (ctypes.c_char_p, [_c_int]),
    "mvp_device_arch": (_c_int, [_c_int, ctypes.c_char_p, _c_int]),
    "mvp_raydirs_forward": (_c_int, [_c_int] * 3 + [_c_void_p] * 5 + [_c_float] + [_c_void_p] * 3 + [_c_void_p]),
    "mvp_aabb_build": (_c_int, [_c_int] * 2 + [_c_void_p] * 4 + [_c_void_p]),
    # N,H,W,K | raypos,raydir | stepsize | tminmax,nodeaabb,primpos,primrot,primscale | TD,TH,TW |
    # tplate | WD,WH,WW | warp,rayrgba,raysat,rayaux,primlist_count,primlist | primlist_cap | fadescale,fadeexp |
    # diag,stream
    "mvp_march_forward": (_c_int, [_c_int] * 4 + [_c_void_p] * 2 + [_c_float] + [_c_void_p] * 5 + [_c_int] * 3 +
                          [_c_void_p] + [_c_int] * 3 + [_c_void_p] * 6 + [_c_int] + [_c_float] * 2 + [_c_void_p] * 2),
    # ... | tplate | WD,WH,WW | warp,raysat,rayaux,primlist_count,primlist | primlist_cap |
    # grad_rayrgba,g_pos,g_rot,g_scale,g_tplate,g_warp | fadescale,fadeexp | diag,stream
    "mvp_march_backward": (_c_int, [_c_int] * 4 + [_c_void_p] * 2 + [_c_float] + [_c_void_p] * 5 + [_c_int] * 3 +
                           [_c_void_p] + [_c_int] * 3 + [_c_void_p] * 5 + [_c_int] + [_c_void_p] * 6 + [_c_float] * 2 +
                           [_c_void_p] * 2),
}
# N,H,W,K | campos,camrot,focal,princpt,pixelcoords | volradius,stepsize | nodeaabb,primpos,primrot,primscale | TD,TH,TW |
# tplate,rayrgba,raysat,rayaux,primlist_count,primlist | primlist_cap | raypos_out,raydir_out,tminmax_out |
# fadescale,fadeexp | diag,stream
SIGNATURES["mvp_march_forward_cams"] = (_c_int, [_c_int] * 4 + [_c_void_p] * 5 + [_c_float] * 2 + [_c_void_p] * 4 +
                                        [_c_int] * 3 + [_c_void_p] * 6 + [_c_int] + [_c_void_p] * 3 + [_c_float] * 2 +
                                        [_c_void_p] * 2)
# N,H,W,K | raypos,raydir,tminmax | campos,camrot,focal,princpt,pixelcoords | volradius,stepsize |
# nodeaabb,primpos,primrot,primscale | TD,TH,TW | tplate_half,rayrgba | fadescale,fadeexp | diag,stream
SIGNATURES["mvp_march_render_half"] = (_c_int, [_c_int] * 4 + [_c_void_p] * 8 + [_c_float] * 2 + [_c_void_p] * 4 + [_c_int] * 3 +
                                       [_c_void_p] * 2 + [_c_float] * 2 + [_c_void_p] * 2)
SIGNATURES["mvp_template_to_half"] = (_c_int, [ctypes.c_longlong] + [_c_void_p] * 3)
SIGNATURES["mvp_template_assemble_forward_half"] = (_c_int, [_c_int] * 3 + [_c_void_p] * 4)
SIGNATURES["mvp_template_assemble_forward"] = (_c_int, [_c_int] * 3 + [_c_void_p] * 4)
SIGNATURES["mvp_template_assemble_backward"] = (_c_int, [_c_int] * 3 + [_c_void_p] * 5)
# nh,B -> blocks | F,nh,B | tex,opacity,gain | tplate | stream | F,nh,B | tex,opacity,gain,grad_tplate | gtex,gop,partials | stream
SIGNATURES["mvp_template_assemble_frames_blocks"] = (ctypes.c_longlong, [_c_int] * 2)
SIGNATURES["mvp_template_assemble_frames_forward"] = (_c_int, [_c_int] * 3 + [_c_void_p] * 5)
SIGNATURES["mvp_template_assemble_frames_backward"] = (_c_int, [_c_int] * 3 + [_c_void_p] * 8)
# N,K,rw | pos0,sn | rot0,sn | scale0,sn,sk,sc | posres,sn | rotres,sn | scaleres,sn | outputs / gradients | stream
_POSE_IN = [_c_int, _c_int, ctypes.c_float, _c_void_p, ctypes.c_longlong, _c_void_p, ctypes.c_longlong, _c_void_p,
            ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _c_void_p, ctypes.c_longlong, _c_void_p,
            ctypes.c_longlong, _c_void_p, ctypes.c_longlong]
SIGNATURES["mvp_prim_residuals_forward"] = (_c_int, _POSE_IN + [_c_void_p] * 4)
SIGNATURES["mvp_prim_residuals_backward"] = (_c_int, _POSE_IN + [_c_void_p] * 9)
SIGNATURES["mvp_prim_frame_forward"] = (_c_int, [ctypes.c_longlong] + [_c_void_p] * 4)
SIGNATURES["mvp_prim_frame_backward"] = (_c_int, [ctypes.c_longlong] + [_c_void_p] * 6)
# N,H,W | rayrgba | rayrgb,rayalpha | stream     and     N,H,W | g_rgb,g_alpha | g_rgba | stream
SIGNATURES["mvp_rgba_split_forward"] = (_c_int, [_c_int] * 3 + [_c_void_p] * 3 + [_c_void_p])
SIGNATURES["mvp_rgba_split_backward"] = (_c_int, [_c_int] * 3 + [_c_void_p] * 3 + [_c_void_p])
# B,V,T,ny,nx,y0,sy,x0,sx | volradius | geo,idxim,barim | primpos,du,dv | stream
SIGNATURES["mvp_prim_placement_forward"] = (_c_int, [_c_int] * 9 + [_c_float] + [_c_void_p] * 6 + [_c_void_p])
# B,V,T,ny,nx,y0,sy,x0,sx | volradius | idxim,barim | g_primpos,g_du,g_dv | grad_geo | stream
SIGNATURES["mvp_prim_placement_backward"] = (_c_int, [_c_int] * 9 + [_c_float] + [_c_void_p] * 6 + [_c_void_p])
# ntensors | grads(host array of device ptrs), numels(host array) | sqnorm | stream
SIGNATURES["mvp_grads_sanitize_sqnorm"] = (_c_int, [_c_int] + [_c_void_p] * 3 + [_c_void_p])
# ntensors | grads, numels | sqnorm | max_norm | total_norm | stream
SIGNATURES["mvp_grads_clip_scale"] = (_c_int, [_c_int] + [_c_void_p] * 3 + [_c_float] + [_c_void_p] * 2)
# B, HW | samplecoords, bias1, w1pos, wh, bh, w6, b6, acts, x0, out | stream
SIGNATURES["mvp_bgmlp_forward"] = (_c_int, [_c_int] * 2 + [_c_void_p] * 10 + [_c_void_p])
# B, HW | grad_out, acts, whT, w6, dz, colsum | stream
SIGNATURES["mvp_bgmlp_backward"] = (_c_int, [_c_int] * 2 + [_c_void_p] * 6 + [_c_void_p])
# N, H, W, K, kind, first_block, count | out, total_blocks
SIGNATURES["mvp_march_block_map"] = (_c_int, [_c_int] * 7 + [_c_void_p] * 2)
# primlist_count, nprims | hist[257] | stream
SIGNATURES["mvp_list_demand"] = (_c_int, [_c_void_p, ctypes.c_longlong, _c_void_p, _c_void_p])
SIGNATURES["mvp_pixel_tail_blocks"] = (_c_int, [_c_int] * 2)
# N,H,W | rayrgba,cw,cb,bg,target | irgbrec,ialpha,l1_partials | stream
SIGNATURES["mvp_pixel_tail_forward"] = (_c_int, [_c_int] * 3 + [_c_void_p] * 8 + [_c_void_p])
# N,H,W | rayrgba,cw,bg,target,irgbrec,g_irgbrec,g_ialpha,g_l1 | grad_rayrgba,grad_bg,cwcb_partials | stream
SIGNATURES["mvp_pixel_tail_backward"] = (_c_int, [_c_int] * 3 + [_c_void_p] * 11 + [_c_void_p])
ABI_VERSION = 17
DIAG_WORDS = 8
DIAG_NAMES = ["frontier_overflow", "list_overflow", "slowpath_packets", "max_list", "packets_hit", "list_entries",
              "candidates"]

_lib = None


def use_library(path=None):
    """Tests / timing experiments only: bind a different build of the SAME ABI (build_variants/...), or go back to the
    product library with path=None.  Takes effect on the next get_lib()."""
    global _lib, LIB_PATH
    LIB_PATH = path or os.path.join(_HERE, "libmvp_gfx950.so")
    _lib = None


def get_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libmvp_gfx950.so is not built (%s). Build it with `python -c \"import __graft_entry__ as g; "
                "g.build()\"` or `python ava-256_amd/build.py`. There is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        if lib.mvp_abi_version() != ABI_VERSION:
            raise RuntimeError("libmvp_gfx950.so ABI version %d != expected %d (stale build?)" %
                               (lib.mvp_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def check(code, what):
    if code != 0:
        msg = get_lib().mvp_error_string(code)
        raise RuntimeError("%s failed: %s (code %d)" % (what, msg.decode() if msg else "?", code))
