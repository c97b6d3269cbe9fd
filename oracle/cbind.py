"""ctypes binding of oracle/liborc.so (test infrastructure; see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile rans_oracle.c -> liborc.so with the committed Makefile."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liborc.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        i32p = ctypes.POINTER(ctypes.c_int32)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        f32p = ctypes.POINTER(ctypes.c_float)
        L.orc_pmf_to_quantized_cdf.argtypes = [f32p, ctypes.c_int, ctypes.c_int,
                                               ctypes.POINTER(ctypes.c_uint32)]
        L.orc_pmf_to_quantized_cdf.restype = ctypes.c_int
        L.orc_rans_encode.argtypes = [i32p, i32p, ctypes.c_int, i32p, ctypes.c_int, i32p, i32p,
                                      u8p, ctypes.c_size_t]
        L.orc_rans_encode.restype = ctypes.c_long
        L.orc_rans_decode.argtypes = [u8p, ctypes.c_size_t, i32p, ctypes.c_int, i32p, ctypes.c_int,
                                      i32p, i32p, i32p]
        L.orc_rans_decode.restype = ctypes.c_long
        L.orc_rans_encode_batch.argtypes = [i32p, ctypes.c_int, ctypes.c_int, i32p, ctypes.c_int,
                                            i32p, i32p, u8p, ctypes.c_size_t, u64p]
        L.orc_rans_encode_batch.restype = ctypes.c_long
        L.orc_rans_decode_batch.argtypes = [u8p, u64p, ctypes.c_int, ctypes.c_int, i32p,
                                            ctypes.c_int, i32p, i32p, i32p]
        L.orc_rans_decode_batch.restype = ctypes.c_int
        L.orc_quantise.argtypes = [f32p, ctypes.c_int, ctypes.c_int, f32p, f32p, f32p, i32p]
        L.orc_quantise.restype = None
        _LIB = L
    return _LIB


def _p(a, ctype):
    return a.ctypes.data_as(ctypes.POINTER(ctype))


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def pmf_to_quantized_cdf(pmf, precision=16):
    pmf = np.ascontiguousarray(pmf, dtype=np.float32)
    out = np.zeros(len(pmf) + 1, dtype=np.uint32)
    rc = lib().orc_pmf_to_quantized_cdf(_p(pmf, ctypes.c_float), len(pmf), precision,
                                        _p(out, ctypes.c_uint32))
    if rc != 0:
        raise ValueError("pmf_to_quantized_cdf: no donor frequency")
    return out


def rans_encode(symbols, cdf, cdf_len, offset, index=None):
    """One image: int32 symbols[n] -> bytes (A13)."""
    symbols, cdf, cdf_len, offset = _i32(symbols), _i32(cdf), _i32(cdf_len), _i32(offset)
    n = symbols.shape[0]
    cap = 8 + 8 * 10 * n
    out = np.empty(cap, dtype=np.uint8)
    idx = None if index is None else _p(_i32(index), ctypes.c_int32)
    k = lib().orc_rans_encode(_p(symbols, ctypes.c_int32), idx, n, _p(cdf, ctypes.c_int32),
                              cdf.shape[1], _p(cdf_len, ctypes.c_int32),
                              _p(offset, ctypes.c_int32), _p(out, ctypes.c_uint8), cap)
    if k < 0:
        raise RuntimeError("orc_rans_encode failed")
    return out[:k].tobytes()


def rans_decode(data, n, cdf, cdf_len, offset, index=None):
    """One image: bytes -> int32 symbols[n] (A14)."""
    cdf, cdf_len, offset = _i32(cdf), _i32(cdf_len), _i32(offset)
    buf = np.frombuffer(data, dtype=np.uint8).copy()
    out = np.empty(n, dtype=np.int32)
    idx = None if index is None else _p(_i32(index), ctypes.c_int32)
    k = lib().orc_rans_decode(_p(buf, ctypes.c_uint8), len(buf), idx, n, _p(cdf, ctypes.c_int32),
                              cdf.shape[1], _p(cdf_len, ctypes.c_int32),
                              _p(offset, ctypes.c_int32), _p(out, ctypes.c_int32))
    if k < 0:
        raise RuntimeError("orc_rans_decode failed")
    return out


def rans_encode_batch(symbols, cdf, cdf_len, offset):
    """symbols int32 [B, C] -> (payload uint8[total], offsets uint64[B+1])."""
    symbols, cdf, cdf_len, offset = _i32(symbols), _i32(cdf), _i32(cdf_len), _i32(offset)
    B, C = symbols.shape
    cap = B * (8 + 8 * 10 * C)
    out = np.empty(cap, dtype=np.uint8)
    off = np.zeros(B + 1, dtype=np.uint64)
    k = lib().orc_rans_encode_batch(_p(symbols, ctypes.c_int32), B, C, _p(cdf, ctypes.c_int32),
                                    cdf.shape[1], _p(cdf_len, ctypes.c_int32),
                                    _p(offset, ctypes.c_int32), _p(out, ctypes.c_uint8), cap,
                                    _p(off, ctypes.c_uint64))
    if k < 0:
        raise RuntimeError("orc_rans_encode_batch failed")
    return out[:k].copy(), off


def rans_decode_batch(payload, off, C, cdf, cdf_len, offset):
    cdf, cdf_len, offset = _i32(cdf), _i32(cdf_len), _i32(offset)
    payload = np.ascontiguousarray(payload, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    B = len(off) - 1
    out = np.empty((B, C), dtype=np.int32)
    rc = lib().orc_rans_decode_batch(_p(payload, ctypes.c_uint8), _p(off, ctypes.c_uint64), B, C,
                                     _p(cdf, ctypes.c_int32), cdf.shape[1],
                                     _p(cdf_len, ctypes.c_int32), _p(offset, ctypes.c_int32),
                                     _p(out, ctypes.c_int32))
    if rc != 0:
        raise RuntimeError("orc_rans_decode_batch failed")
    return out


def quantise(z, bias, exp_scale, median):
    """A4 + round: fp32 z [B, C] -> int32 symbols [B, C]."""
    z = np.ascontiguousarray(z, dtype=np.float32)
    B, C = z.shape
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    bias, exp_scale, median = f(bias), f(exp_scale), f(median)
    out = np.empty((B, C), dtype=np.int32)
    lib().orc_quantise(_p(z, ctypes.c_float), B, C, _p(bias, ctypes.c_float),
                       _p(exp_scale, ctypes.c_float), _p(median, ctypes.c_float),
                       _p(out, ctypes.c_int32))
    return out
