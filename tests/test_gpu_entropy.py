"""GPU parity: the HIP entropy stage, called through the C-ABI, against the CPU oracle.
Integer work => bit-exact."""
import os

import numpy as np
import pytest
import torch

from conftest import BETAS, GOLDEN, load_tables, sample_symbols
from oracle import cbind, container, eb

pytestmark = pytest.mark.gpu


def _dev_tables(tab):
    from lossyless_amd import _lib  # noqa: F401  (loads the HIP library or fails loudly)
    d = {k: torch.from_numpy(np.ascontiguousarray(tab[k])).cuda()
         for k in ("cdf", "cdf_len", "offset", "median", "exp_scale", "bias")}
    d["W"] = int(tab["cdf"].shape[1])
    return d


def _encode_symbols(sym, tab, record_prefix=False):
    """int32 [B,C] -> list[bytes] via lla_rans_encode_batch + lla_rans_compact."""
    from lossyless_amd import _lib
    from lossyless_amd.entropy import EntropyBottleneck
    L = _lib.lib()
    d = _dev_tables(tab)
    B, C = sym.shape
    s = torch.from_numpy(np.ascontiguousarray(sym)).cuda()
    stride = int(L.lla_rans_max_encoded_bytes(C))
    scratch = torch.empty(max(B, 1) * stride, dtype=torch.uint8, device="cuda")
    lengths = torch.empty(max(B, 1), dtype=torch.int32, device="cuda")
    rc = L.lla_rans_encode_batch(_lib.ptr(s), B, C, _lib.ptr(d["cdf"]), d["W"], _lib.ptr(d["cdf_len"]),
                                 _lib.ptr(d["offset"]), _lib.ptr(scratch), stride, _lib.ptr(lengths),
                                 _lib.stream_ptr())
    _lib.check(rc, "lla_rans_encode_batch")
    payload, off = EntropyBottleneck.compact_device(scratch, stride, lengths, B, record_prefix)
    off = off.cpu().numpy()
    blob = payload[: int(off[-1])].cpu().numpy().tobytes()
    return blob, off


@pytest.mark.parametrize("tag", BETAS)
def test_encode_golden_symbols_bit_exact(tag):
    tab = load_tables(tag)
    sym = np.load(os.path.join(GOLDEN, f"symbols_{tag}.npy"))
    blob, off = _encode_symbols(sym, tab)
    for i in range(sym.shape[0]):
        want = cbind.rans_encode(sym[i], tab["cdf"], tab["cdf_len"], tab["offset"])
        assert blob[int(off[i]):int(off[i + 1])] == want, f"image {i}"
    # record-prefixed compaction == body of the reference container
    body, _ = _encode_symbols(sym, tab, record_prefix=True)
    with open(os.path.join(GOLDEN, f"golden_{tag}.bin"), "rb") as f:
        assert f.read() == (len(sym).to_bytes(4, "big") + body)


@pytest.mark.parametrize("B", [1, 63, 64, 65, 257, 1024])
def test_encode_random_batches_match_oracle(B, tables_b005):
    sym = sample_symbols(tables_b005, B, seed=B, escape_boost=0.02)
    blob, off = _encode_symbols(sym, tables_b005)
    pay, ooff = cbind.rans_encode_batch(sym, tables_b005["cdf"], tables_b005["cdf_len"],
                                        tables_b005["offset"])
    assert np.array_equal(off.astype(np.uint64), ooff)
    assert blob == pay.tobytes()


def test_encode_edge_symbols(tables):
    C = tables["cdf"].shape[0]
    rows = [np.full(C, -2 ** 29, np.int32),                 # all escaped, 8-digit payloads (max size)
            np.full(C, 2 ** 29, np.int32),
            tables["offset"].astype(np.int32),              # v = 0 everywhere
            (tables["offset"] + tables["cdf_len"] - 2).astype(np.int32),  # exactly the escape index
            (tables["offset"] + tables["cdf_len"] - 3).astype(np.int32),  # last regular symbol
            (tables["offset"] - 1).astype(np.int32)]        # v = -1
    sym = np.stack(rows)
    blob, off = _encode_symbols(sym, tables)
    for i in range(len(rows)):
        want = cbind.rans_encode(sym[i], tables["cdf"], tables["cdf_len"], tables["offset"])
        assert blob[int(off[i]):int(off[i + 1])] == want, f"row {i}"


def test_empty_batch():
    tab = load_tables("5e-02")
    blob, off = _encode_symbols(np.zeros((0, 512), np.int32), tab)
    assert blob == b"" and off.tolist() == [0]


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_fused_quantise_encode_matches_oracle(dtype, tables_b005):
    """z -> (z + bias) * exp_scale - median -> rint -> rANS, fused on device, vs the oracle."""
    from lossyless_amd.entropy import EntropyBottleneck
    tab = tables_b005
    rng = np.random.default_rng(7)
    z = (rng.standard_normal((300, 512)) * 1.5).astype(np.float32)
    z[0, :8] = [0.5, -0.5, 1.5, 2.5, 1e4, -1e4, 0.0, 3.0]     # ties + far escapes
    zt = torch.from_numpy(z).to(dtype).cuda()
    z_seen = zt.float().cpu().numpy()                          # what the kernel actually reads
    eb_mod = EntropyBottleneck(512)
    payload, off, sym = eb_mod.encode_device(zt, _dev_tables(tab), want_symbols=True)
    want_sym = eb.symbols_of(z_seen, tab)
    assert np.array_equal(sym.cpu().numpy(), want_sym)
    off = off.cpu().numpy()
    blob = payload[: int(off[-1])].cpu().numpy().tobytes()
    pay, ooff = cbind.rans_encode_batch(want_sym, tab["cdf"], tab["cdf_len"], tab["offset"])
    assert blob == pay.tobytes() and np.array_equal(off.astype(np.uint64), ooff)


def test_quantise_kernel_round_half_even_and_separate_rounding(tables_b005):
    from lossyless_amd import _lib
    tab = tables_b005
    rng = np.random.default_rng(3)
    z = (rng.standard_normal((64, 512)) * 2).astype(np.float32)
    # plant exact .5 boundaries: choose z so that (z+b)*s - m is k + 0.5 in fp32
    d = _dev_tables(tab)
    zt = torch.from_numpy(z).cuda()
    sym = torch.empty((64, 512), dtype=torch.int32, device="cuda")
    rc = _lib.lib().lla_quantise(_lib.ptr(zt), _lib.LLA_Z_F32, 64, 512, _lib.ptr(d["bias"]),
                                 _lib.ptr(d["exp_scale"]), _lib.ptr(d["median"]), _lib.ptr(sym),
                                 _lib.stream_ptr())
    _lib.check(rc, "lla_quantise")
    assert np.array_equal(sym.cpu().numpy(), eb.symbols_of(z, tab))


@pytest.mark.parametrize("tag", BETAS)
def test_decode_golden_container(tag):
    """lla_container_index + lla_rans_decode_batch on the committed container."""
    import ctypes
    from lossyless_amd import _lib
    from lossyless_amd.entropy import EntropyBottleneck
    tab = load_tables(tag)
    sym = np.load(os.path.join(GOLDEN, f"symbols_{tag}.npy"))
    blob = np.fromfile(os.path.join(GOLDEN, f"golden_{tag}.bin"), dtype=np.uint8)
    L = _lib.lib()
    n = ctypes.c_uint32()
    off = np.zeros(len(sym) + 1, np.uint64)
    assert L.lla_container_index(blob.ctypes.data_as(ctypes.c_void_p), blob.size,
                                 off.ctypes.data_as(ctypes.c_void_p), off.size, ctypes.byref(n)) == 0
    assert n.value == len(sym)
    body = np.concatenate([blob[4:], np.zeros(8, np.uint8)])
    m = EntropyBottleneck(512)
    got, status = m.decode_device(torch.from_numpy(body).cuda(),
                                  torch.from_numpy(off.astype(np.int64)).cuda(), len(sym),
                                  _dev_tables(tab), record_prefix=True)
    assert int(status.max()) == 0
    assert np.array_equal(got.cpu().numpy(), sym)


def test_decode_flags_truncated_stream(tables_b005):
    from lossyless_amd.entropy import EntropyBottleneck
    sym = sample_symbols(tables_b005, 4, seed=9)
    pay, off = cbind.rans_encode_batch(sym, tables_b005["cdf"], tables_b005["cdf_len"],
                                       tables_b005["offset"])
    off = off.astype(np.int64)
    off2 = off.copy()
    off2[2] -= 40          # image 1 loses its last 40 bytes (and image 2 starts early)
    m = EntropyBottleneck(512)
    body = np.concatenate([pay, np.zeros(8, np.uint8)])
    _, status = m.decode_device(torch.from_numpy(body).cuda(), torch.from_numpy(off2).cuda(), 4,
                                _dev_tables(tables_b005))
    assert status.cpu().numpy()[1] == 1


def test_full_size_round_trip_properties(tables_b005):
    """BASELINE config-2 size (1024 images): decode(encode(s)) == s, lengths word aligned,
    total payload == oracle total, dequantise/represent agree with the oracle."""
    from lossyless_amd import _lib
    from lossyless_amd.entropy import EntropyBottleneck
    tab = tables_b005
    d = _dev_tables(tab)
    rng = np.random.default_rng(1)
    z = (rng.standard_normal((1024, 512)) * 1.2).astype(np.float16)
    zt = torch.from_numpy(z).cuda()
    m = EntropyBottleneck(512)
    payload, off, sym = m.encode_device(zt, d, want_symbols=True, record_prefix=True)
    offn = off.cpu().numpy()
    lens = np.diff(offn) - 4
    assert (lens % 4 == 0).all() and (lens >= 8).all()
    back, status = m.decode_device(payload, off, 1024, d, record_prefix=True)
    assert int(status.max()) == 0 and torch.equal(back, sym)
    want_sym = eb.symbols_of(z.astype(np.float32), tab)
    assert np.array_equal(sym.cpu().numpy(), want_sym)
    _, ooff = cbind.rans_encode_batch(want_sym, tab["cdf"], tab["cdf_len"], tab["offset"])
    assert int(offn[-1]) == int(ooff[-1]) + 4 * 1024
    # dequantise + represent
    L = _lib.lib()
    zh = torch.empty((1024, 512), dtype=torch.float32, device="cuda")
    _lib.check(L.lla_dequantise(_lib.ptr(back), 1024, 512, _lib.ptr(d["bias"]), _lib.ptr(d["exp_scale"]),
                                _lib.ptr(d["median"]), _lib.ptr(zh), _lib.stream_ptr()), "deq")
    assert np.array_equal(zh.cpu().numpy(), eb.dequantise(want_sym, tab))
    zr = torch.empty_like(zh)
    _lib.check(L.lla_represent(_lib.ptr(zt), _lib.LLA_Z_F16, 1024, 512, _lib.ptr(d["bias"]),
                               _lib.ptr(d["exp_scale"]), _lib.ptr(d["median"]), _lib.ptr(zr),
                               _lib.stream_ptr()), "rep")
    assert torch.equal(zr, zh)


def test_argument_validation():
    from lossyless_amd import _lib
    L = _lib.lib()
    assert L.lla_rans_encode_batch(None, 1, 512, None, 32, None, None, None, 4096, None, None) == -1
    t = torch.zeros(16, dtype=torch.int32, device="cuda")
    u = torch.zeros(64, dtype=torch.uint8, device="cuda")
    # stride below the worst-case bound -> LLA_ECAP
    assert L.lla_rans_encode_batch(_lib.ptr(t), 1, 4, _lib.ptr(t), 4, _lib.ptr(t), _lib.ptr(t),
                                   _lib.ptr(u), 8, _lib.ptr(t), None) == -2


@pytest.mark.parametrize("tag", BETAS)
def test_encode_large_random_batch_matches_oracle(tag):
    """20k images x 512 symbols per table: ~10M (state, frequency) pairs through the encoder's
    reciprocal-multiply division; every stream must equal the oracle's integer division."""
    tab = load_tables(tag)
    sym = sample_symbols(tab, 20000, seed=77, escape_boost=0.05)
    blob, off = _encode_symbols(sym, tab)
    pay, ooff = cbind.rans_encode_batch(sym, tab["cdf"], tab["cdf_len"], tab["offset"])
    assert np.array_equal(off.astype(np.uint64), ooff)
    assert blob == pay.tobytes()


def _random_tables(C, W, seed):
    """Random but valid coder tables: per channel a random number of bins (>= 1 regular + the
    tail), random pmf through the oracle's pmf_to_quantized_cdf, random offsets."""
    rng = np.random.default_rng(seed)
    cdf = np.zeros((C, W), np.int32)
    cdf_len = np.zeros(C, np.int32)
    offset = rng.integers(-40, 5, size=C).astype(np.int32)
    for c in range(C):
        n = int(rng.integers(2, W))                 # bins incl. the tail; cdf_len = n + 1 <= W
        p = rng.random(n).astype(np.float32) ** 3 + 1e-4
        p /= p.sum()
        cdf[c, :n + 1] = cbind.pmf_to_quantized_cdf(p, 16).astype(np.int64)
        cdf_len[c] = n + 1
    return dict(cdf=cdf, cdf_len=cdf_len, offset=offset)


@pytest.mark.parametrize("C,W", [(1, 3), (7, 4), (33, 17), (512, 33), (1000, 12), (250, 100)])
def test_random_table_shapes_encode_decode(C, W):
    """Table geometry other than the three shipped checkpoints: one channel, the smallest legal
    row (one symbol + tail), odd channel counts (vector-load tail path), wide rows, many
    channels -- encode == oracle, decode inverts, through the batch entry points."""
    from lossyless_amd import _lib
    from lossyless_amd.entropy import EntropyBottleneck
    tab = _random_tables(C, W, seed=C * 131 + W)
    B = 70
    sym = sample_symbols(tab, B, seed=5, escape_boost=0.05)
    blob, off = _encode_symbols_raw(sym, tab)
    for i in range(B):
        assert blob[int(off[i]):int(off[i + 1])] == cbind.rans_encode(
            sym[i], tab["cdf"], tab["cdf_len"], tab["offset"]), (C, W, i)
    d = {k: torch.from_numpy(np.ascontiguousarray(tab[k])).cuda() for k in ("cdf", "cdf_len", "offset")}
    payload = torch.from_numpy(np.frombuffer(blob + b"\0\0\0\0", dtype=np.uint8).copy()).cuda()
    offs = torch.from_numpy(off.astype(np.int64)).cuda()
    out = torch.empty((B, C), dtype=torch.int32, device="cuda")
    status = torch.zeros(B, dtype=torch.int32, device="cuda")
    rc = _lib.lib().lla_rans_decode_batch(_lib.ptr(payload), _lib.ptr(offs), 0, B, C, _lib.ptr(d["cdf"]), W,
                                          _lib.ptr(d["cdf_len"]), _lib.ptr(d["offset"]), _lib.ptr(out),
                                          _lib.ptr(status), _lib.stream_ptr())
    _lib.check(rc, "lla_rans_decode_batch")
    assert int(status.max()) == 0 and np.array_equal(out.cpu().numpy(), sym)


def _encode_symbols_raw(sym, tab):
    from lossyless_amd import _lib
    from lossyless_amd.entropy import EntropyBottleneck
    L = _lib.lib()
    d = {k: torch.from_numpy(np.ascontiguousarray(tab[k])).cuda() for k in ("cdf", "cdf_len", "offset")}
    B, C = sym.shape
    W = tab["cdf"].shape[1]
    s = torch.from_numpy(np.ascontiguousarray(sym)).cuda()
    stride = int(L.lla_rans_max_encoded_bytes(C))
    scratch = torch.empty(B * stride, dtype=torch.uint8, device="cuda")
    lengths = torch.empty(B, dtype=torch.int32, device="cuda")
    rc = L.lla_rans_encode_batch(_lib.ptr(s), B, C, _lib.ptr(d["cdf"]), W, _lib.ptr(d["cdf_len"]),
                                 _lib.ptr(d["offset"]), _lib.ptr(scratch), stride, _lib.ptr(lengths),
                                 _lib.stream_ptr())
    _lib.check(rc, "lla_rans_encode_batch")
    payload, off = EntropyBottleneck.compact_device(scratch, stride, lengths, B)
    off = off.cpu().numpy()
    return payload[: int(off[-1])].cpu().numpy().tobytes(), off


def test_tables_too_large_for_lds_are_refused():
    """The per-channel entry points keep table + per-channel scalars in LDS (<= 64 KiB): larger
    geometries must be refused with LLA_EINVAL, not silently mis-coded (the indexed entry points
    have no such limit)."""
    from lossyless_amd import _lib
    L = _lib.lib()
    C, W = 1200, 40                                   # 96 KB of u16 alone
    t = torch.zeros(C * W, dtype=torch.int32, device="cuda")
    v = torch.zeros(C, dtype=torch.int32, device="cuda")
    s = torch.zeros(4 * C, dtype=torch.int32, device="cuda")
    stride = int(L.lla_rans_max_encoded_bytes(C))
    scratch = torch.empty(4 * stride, dtype=torch.uint8, device="cuda")
    lengths = torch.empty(4, dtype=torch.int32, device="cuda")
    rc = L.lla_rans_encode_batch(_lib.ptr(s), 4, C, _lib.ptr(t), W, _lib.ptr(v), _lib.ptr(v),
                                 _lib.ptr(scratch), stride, _lib.ptr(lengths), _lib.stream_ptr())
    assert rc == -1
