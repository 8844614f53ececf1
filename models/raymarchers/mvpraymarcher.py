"""`from models.raymarchers.mvpraymarcher import Raymarcher` (models/autoencoder.py:20, utils.py:101 of the
reference) resolves to the gfx950 build."""
from ava256_amd.raymarcher import Raymarcher  # noqa: F401
