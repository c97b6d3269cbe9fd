"""torch.hub entry points, same names and returns as the reference's hubconf.py:22-52:
``clip_compressor_b005 / _b001 / _b01 (device=DEVICE, **kwargs) -> (compressor, transform)``.

The reference downloads ``v1.0/beta{beta:0.0e}_factorized_rate.pt`` from its GitHub
release (hubconf.py:15,23-26).  Here the same state-dicts ship inside the package with
their integer coding tables already frozen (``lossyless_amd/assets``),
so nothing is fetched; the URL is only tried when an asset is missing.
"""
dependencies = ["torch", "numpy"]

import os

import torch

from lossyless_amd import ClipCompressor as _ClipCompressor

PATH = "https://github.com/YannDubs/lossyless/releases/download/v1.0/beta{beta:0.0e}_factorized_rate.pt"
DEVICE = "cuda" if torch.cuda.is_available() else "cpu"
_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lossyless_amd", "assets")


def _state_dict(beta):
    local = os.path.join(_ASSETS, "beta{beta:0.0e}_factorized_rate.pt".format(beta=beta))
    if os.path.exists(local):
        return torch.load(local, map_location="cpu", weights_only=True)
    return torch.hub.load_state_dict_from_url(PATH.format(beta=beta), progress=False)


def _make(beta, device, kwargs):
    compressor = _ClipCompressor(pretrained_state_dict=_state_dict(beta), device=device, **kwargs)
    return compressor, compressor.preprocess


def clip_compressor_b005(device=DEVICE, **kwargs):
    return _make(0.05, device, kwargs)


def clip_compressor_b001(device=DEVICE, **kwargs):
    return _make(0.01, device, kwargs)


def clip_compressor_b01(device=DEVICE, **kwargs):
    return _make(0.1, device, kwargs)


_DOC = """Invariant CLIP compressor with beta={beta:.0e} on MI355X.

    Parameters
    ----------
    device : str
        Device on which to load the model ("cuda").
    clip_weights : path / dict / "synthetic", optional
        CLIP ViT-B/32 visual weights (default: $LOSSYLESS_CLIP_WEIGHTS, else synthetic).

    Return
    ------
    compressor : nn.Module
        `compressor(X)` returns decompressed representations, `compressor.compress(X)` byte
        strings, `compressor.compress_dataset(dataset, file)` / `decompress_dataset(file)`
        work on whole datasets.
    transform : callable
        Resize to (3,224,224), CLIP normalisation, tensor conversion.
    """
clip_compressor_b005.__doc__ = _DOC.format(beta=0.05)
clip_compressor_b001.__doc__ = _DOC.format(beta=0.01)
clip_compressor_b01.__doc__ = _DOC.format(beta=0.1)
