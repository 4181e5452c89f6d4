"""prediff_amd: MI355X-native (gfx950 / CDNA4) sampling engine for PreDiff latent diffusion.

Drop-in for the reference's sampling hot path behind its own Python call conventions
(SURVEY.md §8(b)): ``CuboidTransformerUNet`` / ``AutoencoderKL`` / ``LatentDiffusion`` keep the
reference constructor kwargs, forward signatures and ``state_dict`` schema; their forward passes
run hand-written HIP kernels from ``libprediff_hip.so`` (C ABI in include/prediff_hip.h).
There is no non-HIP execution path in this package.
"""
__version__ = "0.1.0"

from .autoencoder_kl import AutoencoderKL  # noqa: E402,F401
from .cuboid_transformer_unet import CuboidTransformerUNet  # noqa: E402,F401
from .latent_diffusion import LatentDiffusion  # noqa: E402,F401
