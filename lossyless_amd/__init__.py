"""lossyless_amd -- MI355X-native ``compress_dataset`` hot path of YannDubs/lossyless.

CLIP ViT-B/32 image encoder -> factorized EntropyBottleneck -> rANS -> ``.bin``, behind
the reference's ``ClipCompressor`` API (hub/compressor.py), running in hand-written HIP
kernels for gfx950 reached through the C ABI in ``include/lossyless_amd.h``.
"""
from .compressor import ClipCompressor  # noqa: F401
from .entropy import EntropyBottleneck  # noqa: F401
from .clip_vit import VisionTransformer, synthetic_vit_state_dict  # noqa: F401

__all__ = ["ClipCompressor", "EntropyBottleneck", "VisionTransformer", "synthetic_vit_state_dict"]
