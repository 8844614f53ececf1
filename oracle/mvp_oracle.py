"""ctypes/numpy front-end of the CPU checker ``oracle/mvp_oracle.c``.

TEST INFRASTRUCTURE ONLY: importable from ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; the product package never imports this module.

``Oracle("f64")`` is the checker (float64 arithmetic), ``Oracle("f32")`` runs the same code in
float32 (the arithmetic type of the kernels) and is what ``bench.py`` times as the CPU "port".
Reference semantics followed: see the header of ``mvp_oracle.c`` (file:line list).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force=False):
    """Compile liboracle_f64.so / liboracle_f32.so with gcc (oracle/Makefile)."""
    targets = [os.path.join(_HERE, "liboracle_f64.so"), os.path.join(_HERE, "liboracle_f32.so")]
    src = os.path.join(_HERE, "mvp_oracle.c")
    stale = force or any((not os.path.exists(t)) or os.path.getmtime(t) < os.path.getmtime(src) for t in targets)
    if stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "all"])
    return targets


class Oracle:
    def __init__(self, precision="f64"):
        assert precision in ("f64", "f32")
        build()
        self.dtype = np.float64 if precision == "f64" else np.float32
        self.lib = ctypes.CDLL(os.path.join(_HERE, "liboracle_%s.so" % precision))
        assert self.lib.mvpo_sizeof_real() == np.dtype(self.dtype).itemsize
        self._creal = ctypes.c_double if precision == "f64" else ctypes.c_float

    def max_threads(self):
        """omp_get_max_threads() of the library's OpenMP loops (what a timing of this port actually ran on)."""
        return int(self.lib.mvpo_max_threads())

    # -- helpers ------------------------------------------------------------------------------
    def _a(self, x):
        return np.ascontiguousarray(np.asarray(x), dtype=self.dtype)

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(ctypes.c_void_p)

    # -- entry points -------------------------------------------------------------------------
    def raydirs(self, campos, camrot, focal, princpt, pixelcoords, volradius, hw=None):
        campos, camrot, focal, princpt = map(self._a, (campos, camrot, focal, princpt))
        N = campos.shape[0]
        if pixelcoords is None:
            H, W = hw
            pc = None
        else:
            pc = self._a(pixelcoords)
            H, W = pc.shape[1], pc.shape[2]
        raypos = np.empty((N, H, W, 3), self.dtype)
        raydir = np.empty((N, H, W, 3), self.dtype)
        tminmax = np.empty((N, H, W, 2), self.dtype)
        rc = self.lib.mvpo_raydirs(N, H, W, self._p(campos), self._p(camrot), self._p(focal), self._p(princpt),
                                   self._p(pc), self._creal(volradius), self._p(raypos), self._p(raydir),
                                   self._p(tminmax))
        assert rc == 0
        return raypos, raydir, tminmax

    def aabb(self, primpos, primrot, primscale):
        primpos, primrot, primscale = map(self._a, (primpos, primrot, primscale))
        N, K = primpos.shape[:2]
        out = np.empty((N, 2 * K - 1, 2, 3), self.dtype)
        rc = self.lib.mvpo_aabb(N, K, self._p(primpos), self._p(primrot), self._p(primscale), self._p(out))
        assert rc == 0
        return out

    def march_forward(self, raypos, raydir, stepsize, tminmax, primpos, primrot, primscale, template,
                      fadescale=8.0, fadeexp=8.0, maxhitboxes=512, want_raysat=True, nodeaabb=None, warp=None, ray_diagnostics=False,
                      edge_eps=4e-6):
        raypos, raydir, tminmax, primpos, primrot, primscale, template = map(
            self._a, (raypos, raydir, tminmax, primpos, primrot, primscale, template))
        N, H, W = raypos.shape[:3]
        K = primpos.shape[1]
        TD, TH, TW = template.shape[2:5]
        assert template.shape[5] == 4
        if nodeaabb is None:
            nodeaabb = self.aabb(primpos, primrot, primscale)
        nodeaabb = self._a(nodeaabb)
        rgba = np.empty((N, H, W, 4), self.dtype)
        raysat = np.empty((N, H, W, 3), self.dtype) if want_raysat else None
        stats = np.zeros(8, np.int64)
        WD = WH = WW = 0
        if warp is not None:
            warp = self._a(warp)
            WD, WH, WW = warp.shape[2:5]
            assert warp.shape[5] == 3
        margin = hitcount = nsamples = None
        if ray_diagnostics:
            margin = np.empty((N, H, W), self.dtype)
            hitcount = np.empty((N, H, W), np.int32)
            nsamples = np.empty((N, H, W), np.int32)
            self.lib.mvpo_set_ray_diagnostics(self._p(margin), self._p(hitcount), self._p(nsamples))
            edge = np.empty((N, H, W), self.dtype)
            self.lib.mvpo_set_edge_diagnostics(self._p(edge), self._creal(edge_eps))
        rc = self.lib.mvpo_march_forward(
            N, H, W, K, self._p(raypos), self._p(raydir), self._creal(stepsize), self._p(tminmax),
            self._p(nodeaabb), self._p(primpos), self._p(primrot), self._p(primscale), TD, TH, TW,
            self._p(template), WD, WH, WW, self._p(warp), self._p(rgba), self._p(raysat), self._creal(fadescale),
            self._creal(fadeexp), int(maxhitboxes), stats.ctypes.data_as(ctypes.c_void_p))
        if ray_diagnostics:
            self.lib.mvpo_set_ray_diagnostics(None, None, None)
            self.lib.mvpo_set_edge_diagnostics(None, self._creal(0.0))
        assert rc == 0
        names = ["rays_hit", "list_len_sum", "samples", "list_overflow", "steps", "rays_saturated"]
        st = dict(zip(names, stats[:6].tolist()))
        if ray_diagnostics:  # per-ray: saturation margin min|alpha_after - 1|, listed primitives, evaluated samples
            # edge: largest opacity increment among the inclusion decisions (box face, march bound) that came within
            # edge_eps world units of flipping (mvp_oracle.c, mvpo_set_edge_diagnostics)
            st.update(margin=margin, hitcount=hitcount, nsamples=nsamples, edge=edge)
        return rgba, raysat, st

    def march_backward(self, raypos, raydir, stepsize, tminmax, primpos, primrot, primscale, template, raysat,
                       grad_rayrgba, fadescale=8.0, fadeexp=8.0, maxhitboxes=512, nodeaabb=None, warp=None):
        """Returns (grad_primpos, grad_primrot, grad_primscale, grad_template) and, with a warp field, grad_warp as a
        fifth element."""
        (raypos, raydir, tminmax, primpos, primrot, primscale, template, raysat, grad_rayrgba) = map(
            self._a, (raypos, raydir, tminmax, primpos, primrot, primscale, template, raysat, grad_rayrgba))
        N, H, W = raypos.shape[:3]
        K = primpos.shape[1]
        TD, TH, TW = template.shape[2:5]
        if nodeaabb is None:
            nodeaabb = self.aabb(primpos, primrot, primscale)
        nodeaabb = self._a(nodeaabb)
        gpos = np.zeros_like(primpos)
        grot = np.zeros_like(primrot)
        gscale = np.zeros_like(primscale)
        gtpl = np.zeros_like(template)
        WD = WH = WW = 0
        gwarp = None
        if warp is not None:
            warp = self._a(warp)
            WD, WH, WW = warp.shape[2:5]
            gwarp = np.zeros_like(warp)
        rc = self.lib.mvpo_march_backward(
            N, H, W, K, self._p(raypos), self._p(raydir), self._creal(stepsize), self._p(tminmax),
            self._p(nodeaabb), self._p(primpos), self._p(primrot), self._p(primscale), TD, TH, TW,
            self._p(template), WD, WH, WW, self._p(warp), self._p(raysat), self._p(grad_rayrgba), self._p(gpos),
            self._p(grot), self._p(gscale), self._p(gtpl), self._p(gwarp), self._creal(fadescale),
            self._creal(fadeexp), int(maxhitboxes))
        assert rc == 0
        if warp is not None:
            return gpos, grot, gscale, gtpl, gwarp
        return gpos, grot, gscale, gtpl
