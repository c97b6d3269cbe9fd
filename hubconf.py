"""torch.hub entry points with the reference's factory names and return shape
(reference hubconf.py:22-52): ``clip_compressor_b005``, ``clip_compressor_b001`` and
``clip_compressor_b01``, each ``(device=..., **kwargs) -> (compressor, transform)``.

The reference fetches one state-dict per rate point from its GitHub release
(hubconf.py:15,23-26).  Here the three state-dicts ship inside the package with their integer
coding tables already frozen (``lossyless_amd/assets``), so nothing is
fetched; the release URL is only a fallback for a missing asset.
"""
dependencies = ["torch", "numpy"]

import pathlib

import torch

from lossyless_amd import ClipCompressor as _Compressor

_ASSET_DIR = pathlib.Path(__file__).resolve().parent / "lossyless_amd" / "assets"
_ASSET_NAME = "beta{beta:0.0e}_factorized_rate.pt"
_RELEASE = "https://github.com/YannDubs/lossyless/releases/download/v1.0/" + _ASSET_NAME
_ON = "cuda" if torch.cuda.is_available() else "cpu"

_DOC = """CLIP ViT-B/32 + entropy bottleneck compressor at beta = {beta:.0e}, MI355X kernels.

    device       : "cuda" (coding runs on the GPU only; there is no CPU fallback)
    clip_weights : path to OpenAI ViT-B-32.pt / a visual state-dict; default $LOSSYLESS_CLIP_WEIGHTS.
                   REQUIRED one way or the other (ValueError otherwise: the rate models only make
                   sense on real CLIP features); "synthetic" = seed-1 random weights, tests/bench only
    gpu_preprocess : False (default) -> ``transform`` is the reference's PIL chain (resize to 224, centre crop,
                   CLIP normalisation, on the host per image); True -> ``transform`` only hands the raw RGB
                   pixels over and the same chain runs on the GPU (bit-identical) inside
                   ``compress_dataset`` / ``compressor(X)`` -- the unchanged reference call
                   ``STL10(transform=transform)`` + ``compress_dataset(dataset, ...)`` then runs at GPU speed
    other kwargs : forwarded to ``lossyless_amd.ClipCompressor``

    Returns ``(compressor, transform)``: ``compressor(X)`` -> reconstructed representations,
    ``compressor.compress(X)`` -> byte strings, ``compress_dataset`` / ``decompress_dataset`` for
    whole datasets (the reference's ``.bin`` format); ``transform`` = resize to 224, centre crop,
    CLIP normalisation.
    """


def _weights_for(beta):
    asset = _ASSET_DIR / _ASSET_NAME.format(beta=beta)
    if asset.exists():
        return torch.load(asset, map_location="cpu", weights_only=True)
    return torch.hub.load_state_dict_from_url(_RELEASE.format(beta=beta), progress=False)


def _entry(tag, beta):
    def factory(device=_ON, **kwargs):
        model = _Compressor(pretrained_state_dict=_weights_for(beta), device=device, **kwargs)
        return model, model.preprocess
    factory.__name__ = factory.__qualname__ = "clip_compressor_" + tag
    factory.__doc__ = _DOC.format(beta=beta)
    return factory


clip_compressor_b005 = _entry("b005", 5e-2)
clip_compressor_b001 = _entry("b001", 1e-2)
clip_compressor_b01 = _entry("b01", 1e-1)
