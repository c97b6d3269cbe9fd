"""GPU parity: arbitrary-row rANS (lla_rans_encode_indexed / _decode_indexed), the
GaussianConditional mirror and the hyperprior twin against the CPU oracle.  Integer work =>
bit-exact strings."""
import numpy as np
import pytest
import torch

from oracle import cbind, eb, gc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tables():
    return gc.derive_tables(gc.get_scale_table())


def _dev(tab):
    return {k: torch.from_numpy(np.ascontiguousarray(tab[k])).cuda() for k in ("cdf", "cdf_len", "offset")}


def _encode(sym, idx, tab):
    from lossyless_amd import _lib
    from lossyless_amd.entropy import EntropyBottleneck
    L = _lib.lib()
    d = _dev(tab)
    B, n = sym.shape
    s = torch.from_numpy(np.ascontiguousarray(sym)).cuda()
    i = torch.from_numpy(np.ascontiguousarray(idx)).cuda()
    stride = int(L.lla_rans_max_encoded_bytes(n))
    scratch = torch.empty(max(B, 1) * stride, dtype=torch.uint8, device="cuda")
    lengths = torch.empty(max(B, 1), dtype=torch.int32, device="cuda")
    T, W = tab["cdf"].shape
    rc = L.lla_rans_encode_indexed(_lib.ptr(s), _lib.ptr(i), B, n, _lib.ptr(d["cdf"]), T, W,
                                   _lib.ptr(d["cdf_len"]), _lib.ptr(d["offset"]), _lib.ptr(scratch),
                                   stride, _lib.ptr(lengths), _lib.stream_ptr())
    _lib.check(rc, "lla_rans_encode_indexed")
    payload, off = EntropyBottleneck.compact_device(scratch, stride, lengths, B)
    return payload, off


def _decode(payload, off, idx, tab):
    from lossyless_amd import _lib
    d = _dev(tab)
    B, n = idx.shape
    i = torch.from_numpy(np.ascontiguousarray(idx)).cuda()
    out = torch.empty((B, n), dtype=torch.int32, device="cuda")
    status = torch.zeros(max(B, 1), dtype=torch.int32, device="cuda")
    T, W = tab["cdf"].shape
    rc = _lib.lib().lla_rans_decode_indexed(_lib.ptr(payload), _lib.ptr(off), 0, B, n, _lib.ptr(i),
                                            _lib.ptr(d["cdf"]), T, W, _lib.ptr(d["cdf_len"]),
                                            _lib.ptr(d["offset"]), _lib.ptr(out), _lib.ptr(status),
                                            _lib.stream_ptr())
    _lib.check(rc, "lla_rans_decode_indexed")
    return out.cpu().numpy(), status.cpu().numpy()


def _draw(rng, B, n, tab, spread=1.0):
    idx = rng.integers(0, tab["cdf"].shape[0], size=(B, n)).astype(np.int32)
    sym = np.rint(rng.normal(size=idx.shape) * tab["scale_table"][idx] * spread).astype(np.int32)
    return sym, idx


@pytest.mark.parametrize("B,n", [(1, 1), (3, 4), (5, 7), (70, 512), (300, 33), (2, 4099)])
def test_indexed_encode_decode_bit_exact(tables, B, n):
    rng = np.random.default_rng(B * 1000 + n)
    sym, idx = _draw(rng, B, n, tables, spread=1.3)
    if n >= 7:   # force escapes of every payload length, both signs, into narrow and wide rows
        sym[0, :7] = [2 ** 30, -(2 ** 30), 10 ** 5, -4000, 17, -17, 0]
        idx[0, :7] = [0, 63, 0, 5, 0, 1, 63]
    payload, off = _encode(sym, idx, tables)
    off_np = off.cpu().numpy()
    blob = payload[: int(off_np[-1])].cpu().numpy().tobytes()
    want = gc.compress(sym, idx, tables)
    for b in range(B):
        assert blob[int(off_np[b]):int(off_np[b + 1])] == want[b], b
    got, status = _decode(payload, off, idx, tables)
    assert (status[:B] == 0).all() and np.array_equal(got, sym)


def test_indexed_matches_the_per_channel_entry_points():
    """indexes[b, c] = c over the factorized tables is exactly lla_rans_encode_batch."""
    from conftest import load_tables, sample_symbols
    tab = load_tables("5e-02")
    sym = sample_symbols(tab, 9, seed=4)
    idx = np.tile(np.arange(sym.shape[1], dtype=np.int32), (sym.shape[0], 1))
    payload, off = _encode(sym, idx, tab)
    off_np = off.cpu().numpy()
    blob = payload[: int(off_np[-1])].cpu().numpy().tobytes()
    for b in range(sym.shape[0]):
        assert blob[int(off_np[b]):int(off_np[b + 1])] == cbind.rans_encode(
            sym[b], tab["cdf"], tab["cdf_len"], tab["offset"])
    got, status = _decode(payload, off, idx, tab)
    assert (status == 0).all() and np.array_equal(got, sym)


def test_indexed_argument_checks_and_clamping(tables):
    from lossyless_amd import _lib
    L = _lib.lib()
    assert L.lla_rans_encode_indexed(None, None, 0, 4, None, 1, 3, None, None, None, 0, None, None) == 0
    assert L.lla_rans_encode_indexed(None, None, 2, 4, None, 1, 3, None, None, None, 0, None, None) == -1
    rng = np.random.default_rng(0)
    sym, idx = _draw(rng, 4, 16, tables)
    bad = idx.copy()
    bad[0, 0], bad[1, 3] = -5, 10 ** 6            # out-of-range rows are clamped, not dereferenced
    idx[0, 0], idx[1, 3] = 0, 63
    p1, o1 = _encode(sym, bad, tables)
    p2, o2 = _encode(sym, idx, tables)
    assert torch.equal(o1, o2) and torch.equal(p1[: int(o1[-1])], p2[: int(o2[-1])])
    # truncated stream -> status 1, no crash
    got, status = _decode(p2, torch.clamp(o2 - 4, min=0), idx, tables)
    assert status.max() == 1 or not np.array_equal(got, sym)


def test_gaussian_conditional_compress_matches_oracle(tables):
    from lossyless_amd.entropy import GaussianConditional
    g = GaussianConditional(None).cuda().eval()
    g.update_scale_table(gc.get_scale_table())
    gen = torch.Generator().manual_seed(2)
    B, C = 37, 512
    scales = torch.exp(torch.randn(B, C, 1, 1, generator=gen) * 2)
    means = torch.randn(B, C, 1, 1, generator=gen) * 3
    x = means + torch.randn(B, C, 1, 1, generator=gen) * scales
    idx = g.build_indexes(scales.cuda())
    strings = g.compress(x.cuda(), idx, means=means.cuda())
    want_idx = gc.build_indexes(scales.numpy(), tables["scale_table"])
    assert np.array_equal(idx.cpu().numpy(), want_idx)
    sym = gc.symbols_of(x.numpy(), means.numpy())
    assert strings == gc.compress(sym, want_idx, tables)
    back = g.decompress(strings, idx, means=means.cuda())
    assert back.shape == x.shape
    assert np.array_equal(back.cpu().numpy(), sym.astype(np.float32).reshape(x.shape) + means.numpy())
    # no means; eval-mode forward returns the same values as decompress(compress(.))
    s2 = g.compress(x.cuda(), idx)
    out, lik = g(x.cuda(), scales.cuda())
    assert torch.equal(g.decompress(s2, idx), out) and bool((lik > 0).all())


def test_hyperprior_twin_roundtrip_and_oracle_composition(tables):
    """HRateHyperprior.compress == oracle(EB strings of side_z) + oracle(GC strings of z_in with the
    rows / means derived from the decoded side information); decompress inverts it."""
    from lossyless_amd.rates import HRateHyperprior
    torch.manual_seed(3)
    m = HRateHyperprior(512).cuda().eval()
    with torch.no_grad():
        m.scaling.fill_(0.7)
        m.biasing.normal_(0, 0.1)
    m.update(force=True)
    z = torch.randn(19, 512, generator=torch.Generator().manual_seed(5)).cuda() * 2
    z_strings, side_strings = m.compress(z)
    assert len(z_strings) == len(side_strings) == 19

    # side information: factorized bottleneck over 102 channels, oracle coder on the same symbols
    ebm = m.entropy_bottleneck
    z_in = m.process_z_in(z)
    side_z = m.side_encoder(z_in)
    med = ebm._medians().detach()
    side_sym = torch.round(side_z - med[None, :]).to(torch.int32).cpu().numpy()
    cdf, ln, off = (ebm._quantized_cdf.cpu().numpy(), ebm._cdf_length.cpu().numpy(),
                    ebm._offset.cpu().numpy())
    assert side_strings == [cbind.rans_encode(s, cdf, ln, off) for s in side_sym]

    # conditional part: rows and "means" exactly as rates.py:686-699 derives them
    side_hat = torch.from_numpy(side_sym).float().cuda() + med[None, :]
    scales_hat = m.z_encoder(side_hat).chunk(2, -1)[0]
    idx = gc.build_indexes(scales_hat.detach().cpu().numpy(), tables["scale_table"])
    sym = gc.symbols_of(z_in.detach().cpu().numpy(), scales_hat.detach().cpu().numpy())
    assert z_strings == gc.compress(sym, idx, tables)

    z_hat = m.decompress([z_strings, side_strings])
    want = m.process_z_out(torch.from_numpy(sym).float().cuda() + scales_hat)
    assert z_hat.shape == (19, 512) and torch.equal(z_hat, want)
    assert m.real_rate(z) == 8 * (sum(map(len, z_strings)) + sum(map(len, side_strings))) / 19


def test_committed_gaussian_fixture_on_gpu(tables):
    """The committed strings (tests/golden/gaussian_golden.npz) come out of the HIP encoder and go
    back through the HIP decoder."""
    import os
    from conftest import GOLDEN
    from oracle import container
    g = np.load(os.path.join(GOLDEN, "gaussian_golden.npz"))
    sym, idx = g["symbols"], g["indexes"]
    payload, off = _encode(sym, idx, tables)
    off_np = off.cpu().numpy()
    blob = payload[: int(off_np[-1])].cpu().numpy().tobytes()
    strings = [blob[int(off_np[b]):int(off_np[b + 1])] for b in range(sym.shape[0])]
    assert container.container_bytes(strings) == g["container"].tobytes()
    got, status = _decode(payload, off, idx, tables)
    assert (status == 0).all() and np.array_equal(got, sym)


def test_gemm_f32_matches_an_fp64_linear_and_is_batch_invariant():
    """lla_gemm_f32 (v_mfma_f32_32x32x2_f32): fp32-roundoff close to an fp64 Linear, ragged M / padded N, and
    every output row the same bits whatever batch it is evaluated in."""
    from lossyless_amd import _lib
    g = torch.Generator().manual_seed(0)
    M, N, K = 1000, 104, 512
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()

    def run(x, relu):
        c = torch.empty((x.shape[0], N), dtype=torch.float32, device="cuda")
        rc = _lib.lib().lla_gemm_f32(_lib.ptr(x), K, _lib.ptr(w), K, _lib.ptr(b), _lib.ptr(c), N, x.shape[0], N, K,
                                     int(relu), _lib.stream_ptr(x.device))
        _lib.check(rc, "lla_gemm_f32")
        return c

    for relu in (False, True):
        want = a.double() @ w.double().t() + b.double()
        if relu:
            want = want.clamp_min(0)
        got = run(a, relu)
        assert float((got.double() - want).abs().max()) < 2e-6 * float(want.abs().max() + 1)
    full = run(a, False)
    for lo, hi in ((0, 1), (3, 67), (64, 1000), (999, 1000)):
        assert torch.equal(run(a[lo:hi].contiguous(), False), full[lo:hi])
    assert _lib.lib().lla_gemm_f32(_lib.ptr(a), K, _lib.ptr(w), K, None, _lib.ptr(full), N, M, N, K - 1, 0,
                                   None) == -1     # K % 8


def test_hyperprior_mlps_run_in_fp32_and_agree_with_a_cpu_fp32_evaluation():
    """the reference evaluates side_encoder / z_encoder in fp32 under autocast(False)
    (lossyless/rates.py:104,631-639).  On 1024 x 512 representations the scale indexes computed on the GPU
    (lla_gemm_f32) equal those of a torch-CPU fp32 evaluation of the same network in >= 99.99 % of the positions
    (they differ only where a scale falls within fp32 roundoff of a table boundary); the round-3 fp16 path
    stays available by name and is measurably further away."""
    from lossyless_amd.rates import HRateHyperprior
    torch.manual_seed(11)
    m = HRateHyperprior(512).eval()
    m.update(force=True)
    z = torch.randn(1024, 512, generator=torch.Generator().manual_seed(7)) * 2

    def indexes(mod, zz):
        with torch.no_grad():
            z_in = mod.process_z_in(zz)
            side = mod.side_encoder(z_in)
            med = mod.entropy_bottleneck._medians()
            side_hat = torch.round(side - med) + med
            scales = mod.z_encoder(side_hat).chunk(2, -1)[0]
            return mod.gaussian_conditional.build_indexes(scales).cpu(), side.cpu(), scales.cpu()

    idx_cpu, side_cpu, scales_cpu = indexes(m, z)
    mg = HRateHyperprior(512).eval()
    mg.load_state_dict(m.state_dict())
    mg = mg.cuda()
    idx_gpu, side_gpu, scales_gpu = indexes(mg, z.cuda())
    assert float((side_gpu - side_cpu).abs().max()) < 1e-4
    same = float((idx_gpu == idx_cpu).float().mean())
    print(f"scale indexes equal to the fp32 CPU evaluation: {100 * same:.4f} %")
    assert same >= 0.9999
    mh = HRateHyperprior(512, mlp_precision="fp16").eval()
    mh.load_state_dict(m.state_dict())
    idx_h, _, _ = indexes(mh.cuda(), z.cuda())
    assert float((idx_h == idx_cpu).float().mean()) < same


def test_hyperprior_strings_decode_in_other_batch_sizes():
    """indexes and means must be the same bits on the encoder and on a decoder that evaluates the
    network at another batch size (1024 rows written at once, read back in pieces of <= 128 and one of 9000+)."""
    from lossyless_amd.rates import HRateHyperprior
    torch.manual_seed(5)
    m = HRateHyperprior(512).cuda().eval()
    m.update(force=True)
    z = torch.randn(1024, 512, generator=torch.Generator().manual_seed(9)).cuda() * 2
    z_strings, side_strings = m.compress(z)
    whole = m.decompress([z_strings, side_strings])
    for lo, hi in ((0, 128), (128, 131), (131, 259), (900, 1024)):
        part = m.decompress([z_strings[lo:hi], side_strings[lo:hi]])
        assert torch.equal(part, whole[lo:hi])
    big = m.decompress([z_strings * 9, side_strings * 9])            # 9216 rows in one call
    assert torch.equal(big[:1024], whole) and torch.equal(big[8192:], whole)


def test_evaluator_protocol_on_the_gpu_twins():
    """lossyless/learnable_compressors.py:84,123-177,339-341,436 against both twins on the GPU (the CPU suite runs
    the same stand-in on the host coder: tests/test_twin.py)."""
    import pickle
    from test_twin import _StandInCompressor, _factorized, _model_like_z
    from lossyless_amd.rates import HRateHyperprior
    with torch.no_grad():
        m_cpu = _factorized()
        z = _model_like_z(m_cpu, 96, 3)
        lc_cpu = _StandInCompressor(m_cpu).eval()
        lc_cpu.on_test_epoch_start()
        m = _factorized().cuda()
        lc = _StandInCompressor(m).eval()
        pickle.loads(pickle.dumps(lc))
        _, _, logs0, _ = lc(z.cuda())
        assert set(logs0) == {"H_q_Z", "H_ZlX"}
        lc.on_test_epoch_start()
        z_hat, rates, logs, _ = lc(z.cuda())
        zc, rc, lg, _ = lc_cpu(z)
        assert torch.equal(z_hat.cpu(), zc)                                   # integer stage: bit-exact
        assert abs(float(logs["H_q_Z"]) - float(lg["H_q_Z"])) < 1e-5 * float(lg["H_q_Z"])
        assert logs["n_bits"] == lg["n_bits"]
        assert lc(z.cuda(), is_compress=True) == lc_cpu(z, is_compress=True)  # device coder == host coder
        lc.set_featurizer()
        pickle.loads(pickle.dumps(lc))

        torch.manual_seed(1)
        h = HRateHyperprior(512).cuda()
        lch = _StandInCompressor(h).eval()
        lch.on_test_epoch_start()
        zz = torch.randn(64, 512, generator=torch.Generator().manual_seed(4)).cuda()
        z_hat, rates, logs, other = lch(zz)
        assert {"H_q_ZlS", "H_q_Z", "H_q_S", "H_ZlX", "n_bits", "compress_time", "receiver_time"} <= set(logs)
        assert rates.shape == (64,) and bool(torch.isfinite(rates).all()) and other == {}
        all_strings = lch(zz, is_compress=True)
        assert len(all_strings) == 2 and logs["n_bits"] == 8 * sum(sum(map(len, s)) for s in all_strings) / 64
        lch.set_featurizer()
        pickle.loads(pickle.dumps(lch))
