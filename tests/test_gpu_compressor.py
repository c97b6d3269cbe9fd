"""GPU: the drop-in surface (hubconf factories, ClipCompressor methods, container files)
end to end against the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import load_tables
from oracle import cbind, container, eb

pytestmark = pytest.mark.gpu


def synth_images(B, seed=0):
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073])
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711])
    return ((u8.float() / 255 - mean) / std).half()


@pytest.fixture(scope="module")
def comp():
    import hubconf
    c, _ = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic")
    return c


class _DS(torch.utils.data.Dataset):
    def __init__(self, x, y):
        self.x, self.y = x, y

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        return self.x[i], self.y[i]


def _oracle_strings(comp, x):
    z = comp.clip(x.cuda()).float().cpu().numpy()
    tab = load_tables("5e-02")
    sym = eb.symbols_of(z, tab)
    return [cbind.rans_encode(s, tab["cdf"], tab["cdf_len"], tab["offset"]) for s in sym], sym, tab


def test_compress_bytes_equal_oracle(comp):
    x = synth_images(9).permute(0, 3, 1, 2).contiguous()
    got = comp.compress(x.cuda())
    want, _, _ = _oracle_strings(comp, x)
    assert got == want
    assert abs(comp.get_rate(x.cuda()) - 8 * np.mean([len(s) for s in want])) < 1e-9


def test_forward_equals_decompress_of_compress(comp):
    x = synth_images(6, seed=2).permute(0, 3, 1, 2).contiguous().cuda()
    z_hat = comp(x)
    assert z_hat.shape == (6, 512) and z_hat.dtype == torch.float32
    back = comp.decompress(comp.compress(x))
    assert torch.equal(z_hat, back)
    _, sym, tab = _oracle_strings(comp, x.cpu())
    assert np.array_equal(z_hat.cpu().numpy(), eb.dequantise(sym, tab))


def test_compress_dataset_file_is_the_reference_container(comp, tmp_path, capsys):
    n = 21
    x = synth_images(n, seed=4).permute(0, 3, 1, 2).contiguous().float()
    y = torch.arange(n) % 10
    f, lf = tmp_path / "Z.bin", tmp_path / "Y.npy"
    comp.compress_dataset(_DS(x, y), f, label_file=lf,
                          kwargs_dataloader=dict(batch_size=8, num_workers=0))
    out = capsys.readouterr().out
    assert "Rate:" in out and "bits/img | Encoding:" in out and "img/sec" in out
    want, sym, tab = _oracle_strings(comp, x.half())
    # batches of 8/8/5 go through the tower separately from one batch of 21: per-image
    # results do not depend on the batch, so the file must match the oracle's container
    assert f.read_bytes() == container.container_bytes(want)
    assert np.load(lf).dtype == np.uint16
    Z, Y = comp.decompress_dataset(f, label_file=lf)
    assert Z.shape == (n, 512) and Z.dtype == np.float32 and Y.dtype == np.int64
    assert np.array_equal(Y, y.numpy())
    assert np.array_equal(Z, eb.dequantise(sym, tab))
    assert np.array_equal(Z, comp(x.cuda().half()).cpu().numpy())


def test_tensor_fast_path_nhwc(comp, tmp_path):
    x = synth_images(10, seed=6).cuda()              # NHWC fp16 on device
    f = tmp_path / "Z.bin"
    comp.compress_dataset(x, f, kwargs_dataloader=dict(batch_size=4), is_info=False)
    strings = container.read_container(f)
    z = comp.clip(x).float().cpu().numpy()
    tab = load_tables("5e-02")
    want = [cbind.rans_encode(s, tab["cdf"], tab["cdf_len"], tab["offset"])
            for s in eb.symbols_of(z, tab)]
    assert strings == want


def test_entropy_grouping_does_not_change_the_file(comp, tmp_path):
    """RecordStream codes the embeddings of `entropy_group` tower batches at once; records are
    position-independent, so every grouping (incl. ragged last batch / last group, and a group
    larger than the dataset) gives the same bytes as coding every batch on its own."""
    x = synth_images(23, seed=11).cuda()
    files = {}
    for g in (1, 2, 3, 16):
        f = tmp_path / f"Z{g}.bin"
        comp.compress_dataset(x, f, kwargs_dataloader=dict(batch_size=4), is_info=False,
                              entropy_group=g, coalesce=0)      # one tower pass per batch of 4
        files[g] = f.read_bytes()
    assert files[1] == files[2] == files[3] == files[16]
    for co in (1024, 8, 5):      # small batches gathered into tower batches of `co` images (5: batches get split)
        f = tmp_path / f"Zc{co}.bin"
        comp.compress_dataset(x, f, kwargs_dataloader=dict(batch_size=4), is_info=False, entropy_group=2,
                              coalesce=co)
        assert f.read_bytes() == files[1], co
    per_batch = b"".join(comp.encode_batch_records(x[i:i + 4]).tobytes() for i in range(0, 23, 4))
    assert files[1][4:] == per_batch
    # a stream can be reused after finish(), and an empty stream yields no bytes
    st = comp.record_stream(3)
    assert st.finish().size == 0
    st.push(x[:5]); st.push(x[5:6])
    a = st.finish()
    st.push(x[:6])
    assert np.array_equal(a, st.finish())


def test_record_stream_pipeline_with_growing_and_ragged_batches(comp):
    """The streaming pipeline at sizes where it really runs on the two tower lanes and the coder stream:
    deferred whole-batch passes (700 .. 1024 images), batches that outgrow the embedding buffer (reallocation
    while the other buffer is still being coded), a 33-image straggler, three groups and a reuse after
    finish() -- bytes must equal batch-by-batch coding of the same images (each on the caller's stream when
    < 640 images, split over the lanes and joined otherwise)."""
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(1024, 224, 224, 3, generator=g, device="cuda").half()
    sizes = [700, 1024, 33, 1024, 650, 1000]
    want = b"".join(comp.encode_batch_records(x[:n]).tobytes() for n in sizes)
    for co in (0, 1024):               # one tower pass per push / pushes re-cut into 1024-image tower batches
        st = comp.record_stream(2, coalesce=co)
        for n in sizes:
            st.push(x[:n].clone())      # (clones: the stream must keep its inputs alive itself)
        got = st.finish().tobytes()
        assert got == want, co
        st.push(x[:900].clone())
        assert st.finish().tobytes() == comp.encode_batch_records(x[:900]).tobytes()


def test_short_first_tower_passes_do_not_change_the_bytes(comp, tmp_path):
    """Images that start in host memory go through short first tower passes (``_TOWER_RAMP``: the tower starts after the
    first 1024 images have crossed the bus, not after a whole pass of 8704).  Records are position-independent: the
    stream's bytes are the same for every ramp, whether the pushes are smaller than, equal to or larger than the
    current pass size, and ``compress_dataset`` writes the same file from a host tensor (ramp) as from the same
    tensor on the device (no ramp)."""
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(300, 224, 224, 3, generator=g, device="cuda").half()
    want = comp.encode_batch_records(x).tobytes()
    for ramp, pushes in (((), [300]), ((32, 64), [16] * 18 + [12]), ((32, 64), [32, 64, 204]), ((40, 100), [128, 172]),
                         ((8, 16, 24), [7, 9, 100, 184])):
        st = comp.record_stream(2, 128, ramp)
        pos = 0
        for n in pushes:
            st.push(x[pos:pos + n].clone())
            pos += n
        assert pos == 300 and st.finish().tobytes() == want, (ramp, pushes)
    import lossyless_amd.compressor as cm
    old = cm._TOWER_RAMP
    try:
        cm._TOWER_RAMP = (16, 48)
        fh, fd = tmp_path / "host.bin", tmp_path / "dev.bin"
        comp.compress_dataset(x.cpu(), fh, is_info=False, coalesce=128)      # host tensor (NHWC): ramp 16, 48, then 128
        comp.compress_dataset(x, fd, is_info=False, coalesce=128)                                # device tensor: whole passes
        assert fh.read_bytes() == fd.read_bytes() and fh.read_bytes()[4:] == want
    finally:
        cm._TOWER_RAMP = old


def test_gathered_device_batches_are_read_in_place_and_give_the_same_bytes(comp, monkeypatch):
    """Pushes that are contiguous fp16 device batches of a multiple of 256 images are not copied into a staging batch:
    the tower reads them where they lie, as 256-image blocks (``lla_vit_b32_forward_gather``; 256 images are 49 whole
    256-row tiles of the patch-embedding GEMM).  Same bytes as the copying path (pushes that are not donated) and as
    batch-by-batch coding, for passes that end inside a push, a push larger than a pass, both layouts, and a push the
    tower cannot read in place (300 images) in the middle, which flushes what was gathered."""
    import lossyless_amd.compressor as cm
    g = torch.Generator(device="cuda").manual_seed(12)
    x = torch.randn(1024, 224, 224, 3, generator=g, device="cuda").half()
    calls = []
    real = type(comp.clip).forward_gather
    monkeypatch.setattr(type(comp.clip), "forward_gather", lambda self, blocks, B, out=None: (calls.append((len(blocks), B)), real(self, blocks, B, out=out))[1])
    # (sizes pushed, images per pass, images the tower must have read in place: after the 300-image push the stream is
    # in the middle of a staging batch, and what follows has to be copied behind it to keep the order)
    for sizes, co, in_place in (([512, 256, 1024, 256], 768, 2048), ([1024, 1024], 512, 2048), ([256, 300, 512], 512, 256),
                                ([768], 1024, 768)):
        want = b"".join(comp.encode_batch_records(x[:n]).tobytes() for n in sizes)
        del calls[:]
        st = comp.record_stream(2, co)
        for n in sizes:
            st.push(x[:n], donate=True)
        assert st.finish().tobytes() == want, (sizes, co)
        assert sum(b for _, b in calls) == in_place and all(b <= co for _, b in calls), (sizes, co, calls)
        del calls[:]
        st = comp.record_stream(2, co)      # not donated: everything is copied, nothing is read in place
        for n in sizes:
            st.push(x[:n])
        assert st.finish().tobytes() == want and not calls, ("copying path", sizes, co)
    xc = x[:512].permute(0, 3, 1, 2).contiguous()          # planar layout
    del calls[:]
    st = comp.record_stream(2, 512)
    st.push(xc[:256], donate=True); st.push(xc[256:], donate=True)
    assert st.finish().tobytes() == comp.encode_batch_records(xc).tobytes() and calls == [(2, 512)]


def test_a_pushed_buffer_may_be_refilled_unless_it_was_donated(comp):
    """`push(x)` without `donate=True` must have taken everything it needs from `x` when it
    returns -- the common streaming pattern refills ONE preallocated device buffer between pushes.  512-image pushes
    into passes of 1024: the first push is still waiting for its pass when the buffer is overwritten."""
    g = torch.Generator(device="cuda").manual_seed(21)
    a = torch.randn(512, 224, 224, 3, generator=g, device="cuda").half()
    b = torch.randn(512, 224, 224, 3, generator=g, device="cuda").half()
    want = comp.encode_batch_records(a).tobytes() + comp.encode_batch_records(b).tobytes()
    buf = torch.empty_like(a)
    st = comp.record_stream(2, 1024)
    buf.copy_(a); st.push(buf)
    buf.copy_(b); st.push(buf)            # refilled while the first 512 images wait for their tower pass
    assert st.finish().tobytes() == want and st.gathered_passes == 0
    st = comp.record_stream(2, 1024)      # donated batches (distinct tensors, left alone) are read in place
    st.push(a, donate=True); st.push(b, donate=True)
    assert st.finish().tobytes() == want and st.gathered_passes == 1


def test_record_stream_takes_fp32_and_non_contiguous_device_batches(comp):
    """a pushed CUDA batch that is fp32 or a non-contiguous view is converted INSIDE the stream
    (``_run_tower``), and the converted tensor -- the one the tower reads, possibly after push() has returned -- is
    what the stream keeps alive; with ``coalesce=0`` such batches reach the tower directly.  Same bytes as pushing
    the fp16 contiguous batch, with the allocator given every chance to recycle a freed temporary in between."""
    g = torch.Generator(device="cuda").manual_seed(8)
    x16 = torch.randn(700, 224, 224, 3, generator=g, device="cuda").half()
    want = comp.encode_batch_records(x16).tobytes() + comp.encode_batch_records(x16[:650]).tobytes()
    cases = {"fp32 NHWC": lambda t: t.float(),
             "fp32 NHWC, non-contiguous view": lambda t: torch.cat([t.float(), t.float()], dim=2)[:, :, :224],
             "fp16 NHWC, non-contiguous view": lambda t: torch.cat([t, t], dim=2)[:, :, :224]}
    for name, make in cases.items():
        st = comp.record_stream(2, coalesce=0)
        for n in (700, 650):
            st.push(make(x16[:n]))
            junk = torch.empty_like(x16[:n], dtype=torch.float16).normal_()   # lands in a just-freed block if one exists
            del junk
        assert st.finish().tobytes() == want, name


@pytest.mark.parametrize("name", ["clip_compressor_b01", "clip_compressor_b001"])
def test_other_rate_points(name):
    import hubconf
    c, _ = getattr(hubconf, name)(device="cuda", clip_weights="synthetic")
    x = synth_images(4, seed=1).cuda()
    tag = {"clip_compressor_b01": "1e-01", "clip_compressor_b001": "1e-02"}[name]
    tab = load_tables(tag)
    z = c.clip(x).float().cpu().numpy()
    want = [cbind.rans_encode(s, tab["cdf"], tab["cdf_len"], tab["offset"])
            for s in eb.symbols_of(z, tab)]
    assert c.compress(x) == want
    assert torch.equal(c(x), c.decompress(want))


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


def test_training_side_twin_matches_hub_path(comp):
    """lossyless/rates.py:509-564 twin: same bytes as the hub compressor for the same z."""
    from lossyless_amd.rates import HRateFactorizedPrior
    sd = {k: v for k, v in comp.state_dict().items()}
    m = HRateFactorizedPrior(512, kwargs_ent_bottleneck=dict(init_scale=10, filters=[3, 3, 3, 3]))
    m.load_state_dict(sd)
    m = m.cuda().eval()
    assert m.is_coder_updated
    x = synth_images(5, seed=11).cuda()
    z = comp.clip(x)
    all_strings = m.compress(z)
    assert isinstance(all_strings, list) and len(all_strings) == 1
    assert all_strings[0] == comp.compress(x)
    assert torch.equal(m.decompress(all_strings), comp(x))
    assert abs(m.real_rate(z) - comp.get_rate(x)) < 1e-9


def test_lazy_synthetic_dataset_is_shard_independent(comp, tmp_path):
    """SyntheticImages (BASELINE configs[3] generator): same pixels whatever the batch split."""
    from lossyless_amd.compressor import SyntheticImages
    ds = SyntheticImages(23, seed=5)
    a, b = tmp_path / "a.bin", tmp_path / "b.bin"
    comp.compress_dataset(ds, a, kwargs_dataloader=dict(batch_size=23), is_info=False)
    comp.compress_dataset(ds, b, kwargs_dataloader=dict(batch_size=5), is_info=False)
    assert a.read_bytes() == b.read_bytes()
    x = ds.device_batch(0, 23, "cuda")
    assert x.shape == (23, 224, 224, 3) and x.dtype == torch.float16
    assert torch.equal(ds.device_batch(7, 9, "cuda"), x[7:9])
    # the kernel against the defining formula in torch int64 / fp32 ops (CPU), incl. a range past 2^32 elements
    assert torch.equal(x.cpu(), ds.reference_batch(0, 23))
    big = SyntheticImages(1_000_000, seed=0)
    assert torch.equal(big.device_batch(999_998, 1_000_000, "cuda").cpu(), big.reference_batch(999_998, 1_000_000))
    assert 65 < float((x.float() * 0.27 + 0.45).mul(255).std()) < 85   # ~ uniform bytes (std 73.9)


def test_compressor_pickles_and_deepcopies_after_a_forward(comp):
    """a compressor that has run holds a tower handle (ctypes pointer) and a device workspace; copies
    must start without them, work, and give the same bytes -- and the original must keep working."""
    import copy
    import io
    import pickle
    x = synth_images(5, seed=11).cuda()
    want = comp.compress(x)
    assert comp.clip._tower is not None
    clone = pickle.loads(pickle.dumps(comp))
    deep = copy.deepcopy(comp)
    buf = io.BytesIO()
    torch.save(comp, buf)
    buf.seek(0)
    loaded = torch.load(buf, weights_only=False)
    for c in (clone, deep, loaded):
        assert c.clip._tower is None
        assert c.compress(x) == want
    del clone, deep, loaded            # three lla_tower_destroy calls, each on its own handle
    assert comp.compress(x) == want
