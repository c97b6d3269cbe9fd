"""ORACLE -- test infrastructure only (see oracle/__init__.py).  PARITY UNPINNED.

CPU restatement of compressai==1.1.5 ``GaussianConditional`` as the reference's hyperprior coder
uses it (lossyless/rates.py:567-569 ``get_scale_table``; :296-299 ``update_scale_table``; :698
``build_indexes``; :712/:722 ``compress`` / ``decompress`` with ``means``) [3P-recalled: compressai
is not vendored, recipe per entropy_models.py ``GaussianConditional.update`` /
``_standardized_cumulative`` / ``build_indexes`` and ``EntropyModel.compress``].

The standard-normal quantile is evaluated with scipy (as compressai does); the cumulative with
``erfc`` in fp32 (torch CPU) or float64 (math.erfc, for the robustness bound).
"""
import math

import numpy as np
import torch

from . import cbind


def get_scale_table(lo=0.11, hi=256, levels=64):
    """lossyless/rates.py:567-569."""
    return torch.exp(torch.linspace(math.log(lo), math.log(hi), levels))


def _cum32(x):
    return 0.5 * torch.erfc(-(2 ** -0.5) * x)


def derive_tables(scale_table, tail_mass=1e-9, precision="fp32"):
    """-> dict(cdf int32 [T, W], cdf_len [T], offset [T], scale_table fp32 [T])."""
    import scipy.stats
    st = torch.as_tensor(scale_table, dtype=torch.float32)
    mult = -scipy.stats.norm.ppf(tail_mass / 2)
    center = torch.ceil(st * mult).int()
    length = 2 * center + 1
    W = int(length.max()) + 2
    T = st.shape[0]
    cdf = np.zeros((T, W), dtype=np.int32)
    for i in range(T):
        n = int(length[i])
        k = torch.abs(torch.arange(n).int() - center[i]).float()
        if precision == "fp32":
            upper = _cum32((0.5 - k) / st[i])
            lower = _cum32((-0.5 - k) / st[i])
            pmf = (upper - lower).numpy()
            tail = (2 * lower[:1]).numpy()
        else:
            s = float(st[i])
            c = lambda v: 0.5 * math.erfc(-(2 ** -0.5) * v)
            kk = k.double().numpy()
            up = np.array([c((0.5 - v) / s) for v in kk])
            lo_ = np.array([c((-0.5 - v) / s) for v in kk])
            pmf = (up - lo_).astype(np.float32)
            tail = np.array([2 * lo_[0]], dtype=np.float32)
        row = cbind.pmf_to_quantized_cdf(np.concatenate([pmf, tail]).astype(np.float32), 16)
        cdf[i, :n + 2] = row.astype(np.int64)
    return dict(cdf=cdf, cdf_len=(length + 2).numpy().astype(np.int32),
                offset=(-center).numpy().astype(np.int32), scale_table=st.numpy())


def build_indexes(scales, scale_table, scale_bound=0.11):
    """index = (#levels - 1) - #{levels except the last that are >= max(scale, bound)}."""
    s = np.maximum(np.asarray(scales, dtype=np.float32), np.float32(scale_bound))
    st = np.asarray(scale_table, dtype=np.float32)
    idx = np.full(s.shape, len(st) - 1, dtype=np.int32)
    for level in st[:-1]:
        idx -= (s <= level).astype(np.int32)
    return idx


def symbols_of(values, means=None):
    """EntropyModel.quantize(..., "symbols", means): round-half-even of (values - means)."""
    v = np.asarray(values, dtype=np.float32)
    if means is not None:
        v = v - np.asarray(means, dtype=np.float32)
    return np.rint(v).astype(np.int32)


def compress(symbols, indexes, tab):
    """One string per row (EntropyModel.compress loop -> encode_with_indexes)."""
    symbols = np.asarray(symbols, dtype=np.int32).reshape(len(symbols), -1)
    indexes = np.asarray(indexes, dtype=np.int32).reshape(len(indexes), -1)
    return [cbind.rans_encode(s, tab["cdf"], tab["cdf_len"], tab["offset"], index=i)
            for s, i in zip(symbols, indexes)]


def decompress(strings, indexes, tab):
    indexes = np.asarray(indexes, dtype=np.int32).reshape(len(indexes), -1)
    return np.stack([cbind.rans_decode(s, i.shape[0], tab["cdf"], tab["cdf_len"], tab["offset"], index=i)
                     for s, i in zip(strings, indexes)])
