"""ava-256_amd -- MI355X-native (gfx950 / CDNA4) MVP-raymarch hot path behind ava-256's operator API.

Only what the path needs lives here:
  csrc/            hand-written HIP kernels + the C-ABI (include/mvp_abi.h) -> libmvp_gfx950.so
  build.py         hipcc driver (in-tree build, no torch extension machinery)
  _lib.py          ctypes binding of the C-ABI (fails loudly when the library is missing)
  raydirs.py       compute_raydirs / ComputeRaydirs          (reference: extensions/utils/utils.py:21-51)
  mvpraymarch.py   mvpraymarch / MVPRaymarch / build_accel   (reference: extensions/mvpraymarch/mvpraymarch.py:21-390)
  raymarcher.py    Raymarcher nn.Module                      (reference: models/raymarchers/mvpraymarcher.py:17-54)
  native_shim.py   `mvpraymarchlib` / `utilslib` with the reference's positional signatures (mvpraymarch.cpp:146-405,
                   utils.cpp:46-137) over the C ABI: the reference's unmodified Python glue can bind to this build
  halfslab.py      opt-in RENDER path over fp16 RGBA slabs (round 5): template_to_half, assemble_template_half, render_half*
  assemble.py      fused decoder -> raymarch template assembly  (rgb.py:137-143, assembler.py:261)
  placement.py     primitive placement on the mesh, 3 texels per primitive (row N2; assembler.py:118-122,143-206)
  gradclip.py      multi-tensor NaN/Inf masking + gradient clipping (row N4; ddp-train.py:434-441)
  trainloop.py     the reference-shaped optimisation loop around the operators (ddp-train.py:362-442), stand-in decoder
  dist_util.py     one-process-per-GPU helpers (camera sharding, RCCL/gloo init)
  scene.py         seeded synthetic scenes for tests and bench

The directory name contains a hyphen (it is the name the build contract asks for); import it as
``ava256_amd`` (a two-line alias package next to it).  There is NO CPU fallback: every operator raises
if its tensors are not on a HIP device or if libmvp_gfx950.so has not been built.
"""
from .raydirs import ComputeRaydirs, compute_raydirs  # noqa: F401
from .mvpraymarch import MVPRaymarch, MVPRaymarchFromCameras, build_accel, mvpraymarch, mvpraymarch_from_cameras  # noqa: F401
from .raymarcher import Raymarcher  # noqa: F401

__all__ = ["compute_raydirs", "ComputeRaydirs", "mvpraymarch", "MVPRaymarch", "build_accel", "Raymarcher",
           "mvpraymarch_from_cameras", "MVPRaymarchFromCameras"]
