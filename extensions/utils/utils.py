"""`from extensions.utils.utils import compute_raydirs` (models/autoencoder.py:19 of the reference) resolves to
the gfx950 build."""
from ava256_amd.raydirs import ComputeRaydirs, compute_raydirs  # noqa: F401
