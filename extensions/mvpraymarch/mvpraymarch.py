"""`from extensions.mvpraymarch.mvpraymarch import mvpraymarch` (models/raymarchers/mvpraymarcher.py:14 of the
reference) resolves to the gfx950 build."""
from ava256_amd.mvpraymarch import MVPRaymarch, build_accel, mvpraymarch  # noqa: F401
