"""segan_pytorch_amd — the SEGAN+/WSEGAN GAN training step, MI355X-native.

Hand-written gfx950 HIP kernels (``csrc/``, built into ``libsegan_hip.so``) behind the
reference's own Python surface (``segan.models.Generator`` / ``Discriminator`` /
``SEGAN``): see DESIGN.md for the path and its boundary.
"""
from . import layout  # noqa: F401  (pure python, no device needed)

__version__ = '0.1.0'
