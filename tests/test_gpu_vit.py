"""GPU parity: CLIP ViT-B/32 kernels, called through the C-ABI, against fp32 references.
Floating point => tolerance, stated per test.  Headline: embeddings within 1e-3 relative
(L2, per image) of the fp32 CPU tower (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import vit as ovit

pytestmark = pytest.mark.gpu


def synth_images(B, seed=0):
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073])
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711])
    return ((u8.float() / 255 - mean) / std).half()  # NHWC fp16


def _gemm(A, W, bias, C, epi):
    from lossyless_amd import _lib
    M, K = A.shape
    N = W.shape[0]
    rc = _lib.lib().lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi,
                                 _lib.stream_ptr())
    _lib.check(rc, "lla_gemm_f16")
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 768), (1000, 768, 3072), (77, 512, 768),
                                   (12800, 2304, 768), (17001, 2304, 768), (12837, 768, 3072),
                                   # round 6, the eight-wave kernel's edges: the shortest K it takes (two K-tiles: first and
                                   # last K-tile of a tile are neighbours), one column tile, the widest N, a last row tile of
                                   # ONE row, fewer tiles than CUs
                                   (9001, 256, 128), (9217, 3072, 192), (9000, 512, 1024)])
def test_gemm_f16_against_fp64(M, N, K):
    """Asymmetric random operands (catches transposed fragments); fp32 accumulate =>
    error bounded by fp16 output rounding: |err| <= 2^-10 |ref| + 1e-3."""
    from lossyless_amd import _lib
    g = torch.Generator().manual_seed(M * 7 + N)
    A = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    ref = (A.double() @ W.double().t() + bias.double())
    C = torch.empty(M, N, dtype=torch.float16, device="cuda")
    _gemm(A, W, bias, C, _lib.LLA_EPI_F16)
    err = (C.double() - ref).abs()
    assert bool((err <= ref.abs() * 2 ** -10 + 1e-3).all()), float(err.max())
    # QuickGELU epilogue
    _gemm(A, W, bias, C, _lib.LLA_EPI_QUICKGELU_F16)
    refg = ref * torch.sigmoid(1.702 * ref)
    err = (C.double() - refg).abs()
    assert bool((err <= refg.abs() * 2 ** -10 + 2e-3).all()), float(err.max())
    # residual epilogue (fp32, in place)
    X0 = torch.randn(M, N, generator=g).cuda()
    X = X0.clone()
    _gemm(A, W, bias, X, _lib.LLA_EPI_RESID_F32)
    err = (X.double() - (X0.double() + ref)).abs()
    assert bool((err <= 1e-4 * (1 + ref.abs())).all()), float(err.max())
    # no bias
    _gemm(A, W, None, C, _lib.LLA_EPI_F16)
    ref0 = A.double() @ W.double().t()
    assert bool(((C.double() - ref0).abs() <= ref0.abs() * 2 ** -10 + 1e-3).all())


def test_gemm_rejects_bad_shapes():
    from lossyless_amd import _lib
    t = torch.zeros(128 * 128, dtype=torch.float16, device="cuda")
    L = _lib.lib()
    assert L.lla_gemm_f16(_lib.ptr(t), _lib.ptr(t), None, _lib.ptr(t), 128, 100, 64, 0, None) == -1
    assert L.lla_gemm_f16(_lib.ptr(t), _lib.ptr(t), None, _lib.ptr(t), 128, 128, 60, 0, None) == -1


def test_layernorm768():
    from lossyless_amd import _lib
    g = torch.Generator().manual_seed(0)
    for rows, stride in [(1, 768), (5, 768), (1000, 768), (7, 50 * 768)]:
        x = (torch.randn(rows * stride, generator=g) * 3 + 1).cuda()
        w = torch.randn(768, generator=g).cuda()
        b = torch.randn(768, generator=g).cuda()
        y = torch.empty(rows, 768, dtype=torch.float16, device="cuda")
        rc = _lib.lib().lla_layernorm768(_lib.ptr(x), stride, _lib.ptr(w), _lib.ptr(b), _lib.ptr(y),
                                         rows, _lib.stream_ptr())
        _lib.check(rc, "ln")
        xs = x.view(rows, stride)[:, :768].double()
        ref = torch.nn.functional.layer_norm(xs, (768,), w.double(), b.double(), 1e-5)
        err = (y.double() - ref).abs()
        assert bool((err <= ref.abs() * 2 ** -10 + 1e-3).all()), float(err.max())


@pytest.mark.parametrize("B", [1, 3, 64])
def test_attention50(B):
    from lossyless_amd import _lib
    g = torch.Generator().manual_seed(B)
    qkv = (torch.randn(B * 50, 2304, generator=g)).half().cuda()
    qkv[:, :768] *= 2.0       # sharper softmax than uniform
    o = torch.zeros(B * 50, 768, dtype=torch.float16, device="cuda")
    rc = _lib.lib().lla_attention50(_lib.ptr(qkv), _lib.ptr(o), B, _lib.stream_ptr())
    _lib.check(rc, "attn")
    q, k, v = qkv.double().view(B, 50, 3, 12, 64).permute(2, 0, 3, 1, 4)  # [B,12,50,64] each
    att = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1)
    ref = (att @ v).permute(0, 2, 1, 3).reshape(B * 50, 768)
    err = (o.double() - ref).abs()
    # P is rounded to fp16 before the second MFMA: abs error ~ 2^-11 * max|v|
    assert float(err.max()) < 4e-3, float(err.max())
    assert float((err / (ref.abs() + 0.05)).mean()) < 2e-3


_TOWERS = {}


def _tower(sd=None, chunk=0):
    """The seed-1 synthetic tower is built once per slice size and shared by the tests of this session (2.3 s of weight
    generation + packing each: the driver's GPU step has a time limit); towers on other weights are built per call."""
    from lossyless_amd.clip_vit import VisionTransformer, synthetic_vit_state_dict
    if sd is not None:
        return VisionTransformer(sd, chunk=chunk).cuda()
    if chunk not in _TOWERS:
        _TOWERS[chunk] = VisionTransformer(synthetic_vit_state_dict(1), chunk=chunk).cuda()
    return _TOWERS[chunk]


def _rel(z, ref):
    return (np.linalg.norm(z - ref, axis=1) / np.linalg.norm(ref, axis=1))


def test_tower_matches_committed_golden_and_fp32_oracle():
    """Both input layouts, vs the committed fp32 vector and vs the oracle run here."""
    x_nhwc = synth_images(4)
    x_nchw = x_nhwc.permute(0, 3, 1, 2).contiguous()
    want = np.load(os.path.join(GOLDEN, "vit_synth_z.npy"))
    tower = _tower()
    z1 = tower(x_nchw.cuda()).float().cpu().numpy()
    z2 = tower(x_nhwc.cuda()).float().cpu().numpy()
    assert _rel(z1, want).max() < 1e-3, _rel(z1, want)
    assert _rel(z2, want).max() < 1e-3, _rel(z2, want)
    from lossyless_amd.clip_vit import synthetic_vit_state_dict
    ref = ovit.vit_b32_forward(synthetic_vit_state_dict(1), x_nchw.float()).numpy()
    assert _rel(ref, want).max() < 1e-5          # oracle reproduces its own fixture here


def test_tower_ragged_batch_and_chunking_are_consistent():
    """B not a multiple of anything; chunked (2 images per slice) == unchunked, bit for bit
    (per-image work does not depend on its neighbours)."""
    x = synth_images(5, seed=3).cuda()
    z_full = _tower(chunk=0)(x)
    z_chunk = _tower(chunk=2)(x)
    assert torch.equal(z_full, z_chunk)
    z_single = torch.cat([_tower(chunk=0)(x[i:i + 1]) for i in range(5)])
    assert torch.equal(z_full, z_single)


def test_two_lane_passes_equal_the_single_stream_pass():
    """The tower-handle entry points (include/lossyless_amd.h, lla_vit_b32_forward_lanes / lla_tower_join):
    joined and deferred passes, ragged sizes, a non-default current stream, a small pass while deferred ones
    are queued -- all must give the bits of plain passes over slices of 250 images.  With the default ONE
    stream this exercises the ordering logic; the same test under LLA_VIT_STREAMS=2 (two lanes, opt-in) is
    tests/test_gpu_variants.py::two_lanes*."""
    tower = _tower()
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(1301, 224, 224, 3, generator=g, device="cuda").half()
    ref = torch.cat([tower(x[i:i + 250]) for i in range(0, 1301, 250)])
    assert torch.equal(tower(x), ref)                       # 2 lanes x slices of 651 / 650
    assert torch.equal(tower(x[:777]), ref[:777])           # odd halves
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        z_side = tower(x[:1024])
        total = z_side.float().sum()                        # consumer on the caller's stream: ordered by the join
    side.synchronize()
    assert torch.equal(z_side, ref[:1024]) and torch.isfinite(total)
    # deferred: three batches queued back to back (whole batches alternate between the lanes), one join
    outs = [torch.zeros(n, 512, dtype=torch.float16, device="cuda") for n in (1024, 277, 700)]
    starts = (0, 1024, 300)
    for o, s0 in zip(outs, starts):
        tower(x[s0:s0 + o.shape[0]], out=o, deferred=True)
    tower.join()
    torch.cuda.synchronize()
    for o, s0 in zip(outs, starts):
        assert torch.equal(o, ref[s0:s0 + o.shape[0]])
    assert torch.equal(tower(x[:900]), ref[:900])           # a joined pass right after deferred ones
    # a SMALL pass (caller's stream, lane 0's buffers) while deferred passes are in flight: must wait for them
    o1 = torch.zeros(1024, 512, dtype=torch.float16, device="cuda")
    o2 = torch.zeros(1024, 512, dtype=torch.float16, device="cuda")
    tower(x[:1024], out=o1, deferred=True)
    tower(x[200:1224], out=o2, deferred=True)
    z_small = tower(x[40:77])
    tower.join()
    torch.cuda.synchronize()
    assert torch.equal(z_small, ref[40:77]) and torch.equal(o1, ref[:1024]) and torch.equal(o2, ref[200:1224])


def test_tower_with_nontrivial_layernorm_and_bias_weights():
    """The synthetic recipe has gamma=1, beta=0; perturb every parameter so that each bias /
    affine path is exercised, then compare with the fp32 oracle."""
    from lossyless_amd.clip_vit import synthetic_vit_state_dict
    sd = synthetic_vit_state_dict(2)
    g = torch.Generator().manual_seed(5)
    for k in sd:
        if k.endswith("weight") and sd[k].dim() == 1:
            sd[k] = 1 + 0.2 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("bias"):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    x = synth_images(3, seed=8)
    ref = ovit.vit_b32_forward(sd, x.permute(0, 3, 1, 2).float()).numpy()
    z = _tower(sd)(x.cuda()).float().cpu().numpy()
    assert _rel(z, ref).max() < 1e-3, _rel(z, ref)


@pytest.mark.parametrize("case", ["outliers", "sharp_attention"])
def test_tower_adds_no_error_beyond_fp16_activations(case):
    """Weights with CLIP-like pathologies: massive residual-stream outliers, or 16x sharper
    attention logits.  The second one is ill-conditioned for ANY fp16-activation tower (1.5e-2 from
    the fp32 result); what is asserted is that the HIP kernels add nothing to the error of an fp32
    evaluation whose activations are merely rounded to fp16 where the tower stores them."""
    from lossyless_amd.clip_vit import synthetic_vit_state_dict
    sd = synthetic_vit_state_dict(3)
    g = torch.Generator().manual_seed(9)
    for k in list(sd):
        if k.endswith("weight") and sd[k].dim() == 1:
            sd[k] = 1 + 0.3 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
        elif sd[k].dim() == 2 and "in_proj" in k and case == "sharp_attention":
            sd[k] = sd[k] * 4.0
    if case == "outliers":
        sd["class_embedding"][[5, 300]] += 100.0
        sd["positional_embedding"][:, [77, 500]] += 50.0
    x = synth_images(4, seed=12)
    xf = x.permute(0, 3, 1, 2).float()
    ref = ovit.vit_b32_forward(sd, xf).numpy()
    emu = _rel(ovit.vit_b32_forward(sd, xf, fp16_storage=True).numpy(), ref)
    hip = _rel(_tower(sd)(x.cuda()).float().cpu().numpy(), ref)
    assert hip.max() <= 1.2 * emu.max() + 1e-4, (hip, emu)
    if case == "outliers":
        assert hip.max() < 1e-3


@pytest.mark.parametrize("logit_gain,mean_peak", [(24.0, 0.66), (36.0, 0.77)])
def test_tower_on_clip_like_statistics(logit_gain, mean_peak):
    """North_star's 1e-3 where it can break: weights with a trained CLIP's pathologies --
    LayerNorm gains around 1 with four channels at 20x, massive activations (|x| up to ~70 on two class-token
    and two every-token channels), per-head query gains that make softmax rows peak at ``mean_peak`` on
    average (0.58-0.81 per layer; N(0, 0.02^2) weights give 0.03) -- ``clip_like_vit_state_dict``.  The HIP
    tower stays within 1e-3 of the fp32 oracle and adds nothing to what fp16 activations alone cost
    (``fp16_storage`` emulation in the oracle)."""
    from lossyless_amd.clip_vit import clip_like_vit_state_dict
    sd = clip_like_vit_state_dict(1, logit_gain=logit_gain)
    x = synth_images(4, seed=12)
    xc = x.permute(0, 3, 1, 2).float()
    ref = ovit.vit_b32_forward(sd, xc).numpy()
    floor = _rel(ovit.vit_b32_forward(sd, xc, fp16_storage=True).numpy(), ref)
    err = _rel(_tower(sd)(x.cuda()).float().cpu().numpy(), ref)
    print(f"clip-like weights, logit gain {logit_gain}: HIP {err.max():.2e}, fp16-activation floor {floor.max():.2e}")
    assert err.max() < 1e-3, (err, floor)
    assert err.max() < 1.3 * floor.max() + 1e-4, (err, floor)


def test_symbol_mismatch_rate_against_the_fp32_tower():
    """SURVEY.md section 7 'hard parts': 1e-3 on the embedding does not decide which side of a rounding boundary
    a value lands on -- report how many of the 512 symbols per image differ between the HIP tower and the fp32
    oracle tower, per rate point (bench.py prints the same figure).  With the shipped scalings (exp(scaling) =
    2.5-4 / 5-6.4 / 25) a 4e-4 relative error moves a symbol with probability ~ error * |z| * exp(scaling)."""
    import bench
    from lossyless_amd.clip_vit import synthetic_vit_state_dict
    x = synth_images(16, seed=3)
    z = _tower()(x.cuda()).float().cpu().numpy()
    ref = ovit.vit_b32_forward(synthetic_vit_state_dict(1), x.permute(0, 3, 1, 2).float()).numpy()
    rates = bench.symbol_mismatch_rates(z, ref)
    assert set(rates) == {"b01", "b005", "b001"}
    for tag, r in rates.items():
        assert 0.0 <= r["rate"] <= 0.2 and r["symbols"] == 16 * 512, rates
    # finer quantisation steps flip more often
    assert rates["b01"]["rate"] <= rates["b005"]["rate"] <= rates["b001"]["rate"], rates
    print(rates)


def test_large_ragged_batches_are_cut_for_the_four_wave_kernel_and_walked_in_both_directions():
    """Round 4: a ragged slice of >= 256 images is cut into a multiple of 128 images (whole 256-row tiles: the
    four-wave GEMM) + the rest (small-M kernels), and the tower's kernels walk the rows in alternating directions
    (GemmParams::rev).  Neither may change a bit: 1301 images in one call == slices of 250 == slices of 37, and the
    embeddings do not depend on where in a batch an image sits."""
    tower = _tower()
    g = torch.Generator(device="cuda").manual_seed(21)
    x = torch.randn(1301, 224, 224, 3, generator=g, device="cuda").half()
    ref = torch.cat([tower(x[i:i + 250]) for i in range(0, 1301, 250)])
    assert torch.equal(tower(x), ref)                                   # 1280 (q4) + 21
    assert torch.equal(tower(x[:1152]), ref[:1152])                     # 9 x 128: no cut
    assert torch.equal(torch.cat([tower(x[i:i + 37]) for i in range(0, 370, 37)]), ref[:370])
    assert torch.equal(tower(x.flip(0)).flip(0), ref)


def test_layernorm_in_the_residual_gemm_epilogues_gives_the_bits_of_the_layernorm_kernel():
    """Round 5: slices of whole 256-row tiles apply ln_2 / the next block's ln_1 in the epilogue of the
    residual GEMM in front of them (gemm_q4.hip EPI_RESID_LNX: the three column tiles of a row tile exchange exact
    per-row partial sums; row tiles whose column tiles miss each other are redone by lnx_cleanup_kernel) instead of
    launching layernorm768_kernel.  One arithmetic everywhere (gemm_common.h ln_finish / ln_affine), so nothing may
    change: default == LayerNorm kernels only == every row tile through the clean-up kernel == every column tile
    waiting as long as it takes == the same images in slices the four-wave kernel does not take.  Non-trivial gamma /
    beta / biases (the synthetic recipe has gamma = 1, beta = 0), and within 1e-3 of the fp32 oracle."""
    from lossyless_amd import _lib
    from lossyless_amd.clip_vit import synthetic_vit_state_dict
    sd = synthetic_vit_state_dict(3)
    g = torch.Generator().manual_seed(9)
    for k in sd:
        if k.endswith("weight") and sd[k].dim() == 1:
            sd[k] = 1 + 0.3 * torch.randn(sd[k].shape, generator=g)
        if k.endswith("bias"):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
    tower = _tower(sd)
    gg = torch.Generator(device="cuda").manual_seed(33)
    x = torch.randn(640, 224, 224, 3, generator=gg, device="cuda").half()      # 32 000 rows = 125 row tiles
    T = _lib.Tower
    z = tower(x)
    h = tower.tower(x.device)
    results = {}
    for name, opts in (("kernels_only", {T.OPT_LNX: 0}), ("all_cleanup", {T.OPT_LNX: 1, T.OPT_LNX_WAIT: -1}),
                       ("wait_for_siblings", {T.OPT_LNX_WAIT: 1 << 24}), ("poll_once", {T.OPT_LNX_WAIT: 0})):
        for k, v in opts.items():
            h.set_option(k, v)
        results[name] = tower(x)
    h.set_option(T.OPT_LNX, 1); h.set_option(T.OPT_LNX_WAIT, 24000)
    for name, zz in results.items():
        assert torch.equal(zz, z), name
    assert torch.equal(torch.cat([tower(x[i:i + 100]) for i in range(0, 300, 100)]), z[:300])   # small-M kernels
    idx = torch.arange(0, 640, 80)
    ref = ovit.vit_b32_forward(sd, x[idx].permute(0, 3, 1, 2).float().cpu()).numpy()
    assert _rel(z[idx].float().cpu().numpy(), ref).max() < 1e-3


def test_slices_beyond_the_four_wave_kernels_32_bit_panel_offsets_fall_back_and_keep_the_bits():
    """a library slice of 14 080+ images (`chunk` is accepted up to 65 536) makes c_proj's A operand
    704 000 x 3072 fp16 >= 4 GiB, which gemm_q4's buffer-descriptor offsets do not address; the LayerNorm-in-epilogue
    path called it without a fallback and the whole forward failed with LLA_EINVAL.  Such slices now take the
    LayerNorm kernels + the ping-pong GEMMs -- and, like every other cut, must not change a bit."""
    from lossyless_amd.clip_vit import VisionTransformer, synthetic_vit_state_dict
    big = VisionTransformer(synthetic_vit_state_dict(1), chunk=14080).cuda().eval()
    ref = _tower()
    g = torch.Generator(device="cuda").manual_seed(77)
    x = torch.randn(14080, 224, 224, 3, generator=g, device="cuda").half()
    z = big(x)
    idx = torch.arange(0, 14080, 110)
    assert torch.equal(z[idx], ref(x[idx].contiguous()))
    assert torch.equal(z[:640], ref(x[:640]))
