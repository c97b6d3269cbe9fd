"""GPU: CLIP RN50 visual tower (lossyless/architectures.py:367-371) -- the
GEMM epilogues it adds, and the whole tower against the fp32 CPU oracle (oracle/rn50.py)."""
import numpy as np
import pytest
import torch

from lossyless_amd import _lib
from oracle import rn50 as orn50

pytestmark = pytest.mark.gpu


def synth_images(B, seed=0):
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073])
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711])
    return ((u8.float() / 255 - mean) / std).half()


@pytest.mark.parametrize("M,N,K,lda,ldc,epi", [(300, 128, 64, 128, 128, 4), (3136, 256, 64, 128, 256, 5),
                                                (77, 128, 320, 320, 128, 4), (1000, 512, 1152, 1152, 512, 5),
                                                (130, 1024, 2048, 100 * 2048, 1024, 0)])
def test_gemm_ex_strided_relu_and_add_relu(M, N, K, lda, ldc, epi):
    """1x1 convolutions are GEMMs over NHWC activations with a channel pitch (lda > K), padded output
    columns (ldc), ReLU / residual-add + ReLU epilogues -- against float64."""
    g = torch.Generator().manual_seed(M + N)
    A = (torch.randn(M, lda, generator=g) * 0.5).half().cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    R = (torch.randn(M, N, generator=g)).half().cuda()
    C = torch.full((M, ldc), 7.0, dtype=torch.float16, device="cuda")
    rc = _lib.lib().lla_gemm_f16_ex(_lib.ptr(A), lda, _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), ldc,
                                    _lib.ptr(R) if epi == 5 else None, N, M, N, K, epi, _lib.stream_ptr())
    assert rc == 0
    ref = A[:, :K].double() @ W.double().t() + bias.double()
    if epi == 5:
        ref = ref + R.double()
    if epi in (4, 5):
        ref = ref.clamp_min(0)
    err = (C[:, :N].double() - ref).abs()
    assert bool((err <= ref.abs() * 2 ** -10 + 2e-3).all()), float(err.max())
    if ldc > N:
        assert bool((C[:, N:] == 7.0).all())          # columns beyond N are not touched
    L = _lib.lib()
    assert L.lla_gemm_f16_ex(_lib.ptr(A), K - 8, _lib.ptr(W), None, _lib.ptr(C), ldc, None, 0, M, N, K, 0, None) == -1
    assert L.lla_gemm_f16_ex(_lib.ptr(A), lda, _lib.ptr(W), None, _lib.ptr(C), ldc, None, 0, M, N, K, 5, None) == -1


@pytest.mark.parametrize("M,N,K,lda,ldc,epi", [(300, 128, 64, 64, 64, 4), (1000, 128, 320, 320, 32, 4),
                                                (777, 256, 128, 128, 160, 5)])
def test_gemm_ex_narrow_output_pitch(M, N, K, lda, ldc, epi):
    """ldc < N (a multiple of 32) with the ReLU epilogues: the GEMM computes all N (padded) columns and stores
    only the first ldc, so that a 32- / 64-channel convolution output keeps a 32- / 64-element row pitch.  The
    output buffer is exactly M x ldc with a guard block behind it."""
    g = torch.Generator().manual_seed(M + ldc)
    A = (torch.randn(M, lda, generator=g) * 0.5).half().cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, ldc, generator=g).half().cuda()
    buf = torch.full((M * ldc + 4096,), 7.0, dtype=torch.float16, device="cuda")
    rc = _lib.lib().lla_gemm_f16_ex(_lib.ptr(A), lda, _lib.ptr(W), _lib.ptr(bias), _lib.ptr(buf), ldc,
                                    _lib.ptr(R) if epi == 5 else None, ldc, M, N, K, epi, _lib.stream_ptr())
    assert rc == 0
    ref = A[:, :K].double() @ W[:ldc].double().t() + bias[:ldc].double()
    if epi == 5:
        ref = ref + R.double()
    ref = ref.clamp_min(0)
    C = buf[:M * ldc].view(M, ldc)
    err = (C.double() - ref).abs()
    assert bool((err <= ref.abs() * 2 ** -10 + 2e-3).all()), float(err.max())
    assert bool((buf[M * ldc:] == 7.0).all())             # nothing behind the last row
    L = _lib.lib()
    assert L.lla_gemm_f16_ex(_lib.ptr(A), lda, _lib.ptr(W), None, _lib.ptr(buf), 48, None, 0, M, N, K, 4, None) == -1
    assert L.lla_gemm_f16_ex(_lib.ptr(A), lda, _lib.ptr(W), None, _lib.ptr(buf), ldc, None, 0, M, N, K, 0, None) == -1


@pytest.fixture(scope="module")
def tower():
    from lossyless_amd.clip_rn50 import ModifiedResNet, synthetic_rn50_state_dict
    sd = synthetic_rn50_state_dict(1)
    return sd, ModifiedResNet(sd, chunk=4).cuda()


def _rel(z, ref):
    return np.linalg.norm(z - ref, axis=1) / np.linalg.norm(ref, axis=1)


def test_rn50_tower_matches_fp32_oracle(tower):
    """Both input layouts vs the fp32 oracle; the error budget of an fp16-activation ResNet-50 is what an
    fp32 evaluation with activations merely rounded to fp16 shows (8e-4 on these weights): the HIP tower
    must not add to it, and stays within 1.5e-3 of the fp32 result."""
    sd, net = tower
    x = synth_images(5, seed=3)
    xc = x.permute(0, 3, 1, 2).float()
    ref = orn50.rn50_forward(sd, xc).numpy()
    emu = _rel(orn50.rn50_forward(sd, xc, fp16_storage=True).numpy(), ref)
    z1 = net(x.cuda()).float().cpu().numpy()
    z2 = net(x.permute(0, 3, 1, 2).contiguous().cuda()).float().cpu().numpy()
    assert z1.shape == (5, 1024) and np.array_equal(z1, z2)
    hip = _rel(z1, ref)
    assert hip.max() < 1.5e-3, hip
    assert hip.max() <= 1.3 * emu.max() + 1e-4, (hip, emu)


def test_rn50_error_floor_of_fp16_storage_exceeds_1e3_and_the_tower_sits_on_it():
    """north_star's 1e-3 is a ViT figure.  A ResNet-50 whose activations are STORED in fp16 -- what the reference
    itself runs (``model.half()``), and what any fp16 tower does -- is already 0.9-1.6e-3 away from the fp32
    evaluation before a single product is rounded (oracle with fp16_storage=True; tools/rn50_err_probe.py: weights
    seed 1: 0.94 / 1.07e-3, seed 2: 1.58 / 1.38e-3).  What can be asked of the HIP tower is that it adds nothing:
    measured 0.92-1.73e-3, i.e. within 1.1x of that floor per weight set."""
    from lossyless_amd.clip_rn50 import ModifiedResNet, synthetic_rn50_state_dict
    floors = []
    for wseed in (1, 2):
        sd = synthetic_rn50_state_dict(wseed)
        net = ModifiedResNet(sd, chunk=8).cuda()
        g = torch.Generator().manual_seed(3)
        x = torch.randn(6, 224, 224, 3, generator=g).half()
        xc = x.permute(0, 3, 1, 2).float()
        ref = orn50.rn50_forward(sd, xc).numpy()
        emu = _rel(orn50.rn50_forward(sd, xc, fp16_storage=True).numpy(), ref)
        hip = _rel(net(x.cuda()).float().cpu().numpy(), ref)
        floors.append(emu.max())
        assert hip.max() <= 1.2 * emu.max() + 1e-4, (wseed, hip, emu)
        assert hip.max() < 2.5e-3
    assert max(floors) > 1e-3, floors      # the floor itself is above north_star's ViT tolerance


def test_rn50_batch_and_chunk_independence(tower):
    """Per-image results do not depend on the batch or the chunking (chunk = 4: 7 images = 4 + 3)."""
    sd, net = tower
    x = synth_images(7, seed=9).cuda()
    z = net(x)
    singles = torch.cat([net(x[i:i + 1]) for i in range(7)])
    assert torch.equal(z, singles)
    from lossyless_amd.clip_rn50 import ModifiedResNet
    assert torch.equal(z, ModifiedResNet(sd, chunk=7).cuda()(x))
    with pytest.raises(RuntimeError):
        net(x.cpu())


def test_rn50_large_chunks_take_the_persistent_kernels_and_agree_with_small_ones(tower):
    """From ~46 images per chunk layer3's, from ~184 layer4's 1x1 convolutions have >= 9000 rows and run on the
    persistent 256-wide kernel with the line-assembling ReLU / add+ReLU epilogue instead of the one-tile-per-
    workgroup kernel: same MFMA order over K, same (acc + bias) + identity -> ReLU -> fp16 per element, so the
    embeddings must be IDENTICAL to the small-chunk ones (and a ragged chunk, 200 = 192 + 8, rides along)."""
    from lossyless_amd.clip_rn50 import ModifiedResNet
    sd, net = tower                                     # chunk = 4: every GEMM below 9000 rows except layer1 / stem
    x = synth_images(200, seed=21).cuda()
    z_small = net(x)
    z_big = ModifiedResNet(sd, chunk=192).cuda()(x)
    assert torch.equal(z_small, z_big)


def test_rn50_state_dict_with_openai_prefix_and_bn_extras(tower):
    """Keys as clip ships them (num_batches_tracked present) fold the same way."""
    from lossyless_amd.clip_rn50 import ModifiedResNet
    sd, net = tower
    sd2 = dict(sd)
    for k in list(sd2):
        if k.endswith("running_var"):
            sd2[k.replace("running_var", "num_batches_tracked")] = torch.tensor(0)
    x = synth_images(2, seed=1).cuda()
    assert torch.equal(net(x), ModifiedResNet(sd2, chunk=2).cuda()(x))


@pytest.mark.parametrize("n,H,W,cin,pitch,cout,ldc", [(2, 13, 9, 64, 64, 128, 128), (3, 7, 7, 128, 192, 128, 256),
                                                      (1, 56, 56, 64, 64, 128, 128), (5, 3, 1, 256, 256, 256, 256),
                                                      (2, 17, 11, 32, 32, 128, 128), (1, 112, 112, 32, 128, 128, 128),
                                                      # (round 6) >= 9000 output pixels: many tiles per XCD, ragged last row tile
                                                      (4, 56, 56, 64, 64, 128, 128), (2, 70, 70, 128, 192, 256, 256),
                                                      (3, 57, 57, 64, 64, 512, 512)])
def test_implicit_conv3x3_relu_against_float64(n, H, W, cin, pitch, cout, ldc):
    """`lla_conv3x3_relu_f16`: the GEMM loader gathers the nine taps itself (no im2col matrix), out-of-image
    taps read zeros, channel pitch on both sides; vs conv2d in float64 -- image borders, images that are one
    pixel wide, row counts that are not a multiple of the 256-row tile."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(n * 100 + H)
    x = (torch.randn(n, H, W, pitch, generator=g) * 0.5).half().cuda()
    w = (torch.randn(cout, 3, 3, cin, generator=g) * 0.05).half().cuda()          # K order (kh, kw, c)
    kpad = (9 * cin + 63) // 64 * 64
    wk = torch.zeros(cout, kpad, dtype=torch.float16, device="cuda")              # rows zero-padded to kpad
    wk[:, :9 * cin] = w.reshape(cout, -1)
    bias = torch.randn(cout, generator=g).cuda()
    out = torch.full((n, H, W, ldc), 7.0, dtype=torch.float16, device="cuda")
    rc = _lib.lib().lla_conv3x3_relu_f16(_lib.ptr(x), n, H, W, pitch, cin, _lib.ptr(wk), _lib.ptr(bias),
                                         _lib.ptr(out), ldc, cout, _lib.stream_ptr())
    assert rc == 0
    ref = F.conv2d(x[..., :cin].permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), bias.double(),
                   padding=1).clamp_min(0).permute(0, 2, 3, 1)
    err = (out[..., :cout].double() - ref).abs()
    assert bool((err <= ref.abs() * 2 ** -10 + 2e-3).all()), float(err.max())
    if ldc > cout:
        assert bool((out[..., cout:] == 7.0).all())      # columns beyond cout are not touched
    L = _lib.lib()
    assert L.lla_conv3x3_relu_f16(_lib.ptr(x), n, H, W, pitch, 96, _lib.ptr(w), _lib.ptr(bias), _lib.ptr(out), ldc,
                                  cout, _lib.stream_ptr()) != 0          # cin % 64


@pytest.mark.parametrize("n,H,W,cin,cout,pitch,ldc,pool", [(2, 16, 24, 32, 32, 32, 32, 0), (3, 112, 112, 32, 64, 32, 64, 0),
                                                           (3, 112, 112, 32, 64, 32, 64, 1), (2, 56, 56, 64, 64, 64, 64, 0),
                                                           (1, 8, 8, 32, 64, 64, 96, 1), (5, 24, 8, 64, 64, 128, 64, 0),
                                                           (70, 16, 16, 32, 32, 32, 32, 0)])
def test_direct_conv3x3_is_the_implicit_gemm_bit_for_bit(n, H, W, cin, cout, pitch, ldc, pool):
    """`lla_conv3x3_direct_relu_f16` (csrc/conv_direct.hip: one 8 x 8 tile per wave, halo in LDS once) against
    `lla_conv3x3_relu_f16` on the same operands: equal bits (same MFMA, same K order, same epilogue arithmetic); with
    pool=1 against the implicit GEMM followed by the tower's 2 x 2 average pool -- ((a + b) + (c + d)) * 0.25 in fp32
    on the fp16-rounded activations, one rounding.  Borders, more tiles than waves, pitches wider than the channels."""
    g = torch.Generator().manual_seed(n * 1000 + H + cin)
    x = (torch.randn(n, H, W, pitch, generator=g) * 0.5).half().cuda()
    kpad = (9 * cin + 63) // 64 * 64
    wk = torch.zeros(128, kpad, dtype=torch.float16, device="cuda")              # rows padded to 128 as the tower's blob
    wk[:cout, :9 * cin] = (torch.randn(cout, 9 * cin, generator=g) * 0.05).half().cuda()
    bias = torch.zeros(128, device="cuda")
    bias[:cout] = torch.randn(cout, generator=g).cuda()
    L = _lib.lib()
    ref = torch.full((n, H, W, ldc), 7.0, dtype=torch.float16, device="cuda")
    assert L.lla_conv3x3_relu_f16(_lib.ptr(x), n, H, W, pitch, cin, _lib.ptr(wk), _lib.ptr(bias), _lib.ptr(ref), ldc,
                                  128, _lib.stream_ptr()) == 0
    if pool:
        r = ref[..., :cout].float()
        want = (((r[:, 0::2, 0::2] + r[:, 0::2, 1::2]) + (r[:, 1::2, 0::2] + r[:, 1::2, 1::2])) * 0.25).half()
        out = torch.full((n, H // 2, W // 2, ldc), 7.0, dtype=torch.float16, device="cuda")
    else:
        want = ref[..., :cout]
        out = torch.full((n, H, W, ldc), 7.0, dtype=torch.float16, device="cuda")
    rc = L.lla_conv3x3_direct_relu_f16(_lib.ptr(x), n, H, W, pitch, cin, _lib.ptr(wk), kpad, _lib.ptr(bias),
                                       _lib.ptr(out), ldc, cout, pool, _lib.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(out[..., :cout], want)
    if ldc > cout:
        assert bool((out[..., cout:] == 7.0).all())      # columns beyond cout are not touched
    # shapes it does not take are refused (the tower then uses the implicit GEMM)
    assert L.lla_conv3x3_direct_relu_f16(_lib.ptr(x), n, H, W, pitch, cin, _lib.ptr(wk), kpad, _lib.ptr(bias),
                                         _lib.ptr(out), ldc, 128, 0, _lib.stream_ptr()) != 0
    assert L.lla_conv3x3_direct_relu_f16(_lib.ptr(x), n, H - 1, W, pitch, cin, _lib.ptr(wk), kpad, _lib.ptr(bias),
                                         _lib.ptr(out), ldc, cout, 0, _lib.stream_ptr()) != 0


@pytest.mark.parametrize("n,H,W,ldc", [(2, 32, 48, 32), (1, 224, 224, 32), (3, 16, 16, 64), (40, 32, 32, 32)])
def test_direct_rgb_stem_convolution_against_float64(n, H, W, ldc):
    """`lla_conv3x3_rgb_s2_relu_f16` (3 -> 32, stride 2, pad 1, pixels of 6 bytes): vs conv2d in float64 -- the left / top
    padding, tiles that touch no border, more tiles than waves.  Equality with the im2col + GEMM path it replaces is the
    tower A/B below."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(n + H)
    x = (torch.randn(n, H, W, 3, generator=g) * 0.7).half().cuda()
    w = (torch.randn(32, 3, 3, 3, generator=g) * 0.2).half().cuda()               # [cout][kh][kw][c]
    wk = torch.zeros(128, 64, dtype=torch.float16, device="cuda")
    wk[:32, :27] = w.reshape(32, 27)
    bias = torch.zeros(128, device="cuda")
    bias[:32] = torch.randn(32, generator=g).cuda()
    out = torch.full((n, H // 2, W // 2, ldc), 7.0, dtype=torch.float16, device="cuda")
    L = _lib.lib()
    assert L.lla_conv3x3_rgb_s2_relu_f16(_lib.ptr(x), n, H, W, _lib.ptr(wk), 64, _lib.ptr(bias), _lib.ptr(out), ldc,
                                         _lib.stream_ptr()) == 0
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), bias[:32].double(), stride=2,
                   padding=1).clamp_min(0).permute(0, 2, 3, 1)
    err = (out[..., :32].double() - ref).abs()
    assert bool((err <= ref.abs() * 2 ** -10 + 2e-3).all()), float(err.max())
    if ldc > 32:
        assert bool((out[..., 32:] == 7.0).all())
    assert L.lla_conv3x3_rgb_s2_relu_f16(_lib.ptr(x), n, H + 8, W, _lib.ptr(wk), 64, _lib.ptr(bias), _lib.ptr(out), ldc,
                                         _lib.stream_ptr()) != 0          # H % 16


def test_direct_convolutions_equal_the_implicit_gemm_tower(tmp_path):
    """The tower with the direct narrow convolutions (default) == the tower with LLA_RN50_DIRECT=0 (implicit GEMMs + the
    stem's separate average pool), bit for bit (with conv3 + downsample as two GEMMs on both sides: the one-GEMM form of
    layer1 needs the direct kernel's exact-width stores).  Then the default tower -- conv3 + downsample of every stage's
    first block as ONE GEMM over [main | block input], which skips the fp16 rounding of the identity -- against that:
    close, and both within the oracle tolerance elsewhere in this file.  Switches are read once per process."""
    import os
    import subprocess
    import sys
    from conftest import ROOT, ablation_env
    script = tmp_path / "r.py"
    script.write_text(r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from lossyless_amd.clip_rn50 import ModifiedResNet, synthetic_rn50_state_dict
net = ModifiedResNet(synthetic_rn50_state_dict(1), chunk=8).cuda()
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.randn(19, 224, 224, 3, generator=g, device="cuda").half()
np.save(sys.argv[2], net(x).cpu().numpy())
''')
    outs = []
    for direct, fuse in (("0", "0"), ("1", "0"), ("1", "1"), ("0", "1")):
        out = tmp_path / f"z{direct}{fuse}.npy"
        r = subprocess.run([sys.executable, str(script), ROOT, str(out)],
                           env=ablation_env(LLA_RN50_DIRECT=direct, LLA_RN50_FUSE_DS=fuse),
                           capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
        outs.append(np.load(out).astype(np.float64))
    assert np.array_equal(outs[0], outs[1])
    scale = np.abs(outs[1]).max()
    assert np.abs(outs[2] - outs[1]).max() <= 2e-3 * scale and not np.array_equal(outs[2], outs[1])
    assert np.abs(outs[3] - outs[1]).max() <= 2e-3 * scale     # (implicit GEMMs: stages 2-4 fused, layer1 not)
    # round 6: layer1's bottlenecks as one kernel each (default) against the three launches per block (LLA_RN50_FUSED_BLOCK=0)
    out = tmp_path / "z_three.npy"
    r = subprocess.run([sys.executable, str(script), ROOT, str(out)], env=ablation_env(LLA_RN50_FUSED_BLOCK="0"),
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    three = np.load(out).astype(np.float64)
    assert np.abs(three - outs[2]).max() <= 2e-3 * scale and not np.array_equal(three, outs[2])


def test_implicit_convolutions_equal_the_im2col_path(tmp_path):
    """The tower with implicit 3x3 GEMMs == the tower with im2col matrices (LLA_RN50_IM2COL=1), bit for bit
    (same K order, same kernel arithmetic).  The switch is read once per process: two interpreters."""
    import os
    import subprocess
    import sys
    from conftest import ROOT, ablation_env
    script = tmp_path / "r.py"
    script.write_text(r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from lossyless_amd.clip_rn50 import ModifiedResNet, synthetic_rn50_state_dict
net = ModifiedResNet(synthetic_rn50_state_dict(1), chunk=4).cuda()
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(9, 224, 224, 3, generator=g, device="cuda").half()
np.save(sys.argv[2], net(x).cpu().numpy())
''')
    outs = []
    for flag in ("0", "1"):
        out = tmp_path / f"z{flag}.npy"
        r = subprocess.run([sys.executable, str(script), ROOT, str(out)], env=ablation_env(LLA_RN50_IM2COL=flag),
                           capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
        outs.append(np.load(out))
    assert np.array_equal(outs[0], outs[1])


def test_rn50_two_lane_pass_equals_small_passes(tower):
    """From 32 images on, lla_rn50_forward alternates slices between the library's two HIP streams; the
    embeddings must be the bits of single-stream passes (here: 8 images at a time)."""
    _, net = tower
    from lossyless_amd.clip_rn50 import ModifiedResNet, synthetic_rn50_state_dict
    big = ModifiedResNet(synthetic_rn50_state_dict(1), chunk=64).cuda()
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(75, 224, 224, 3, generator=g, device="cuda").half()
    ref = torch.cat([net(x[i:i + 8]) for i in range(0, 75, 8)])
    assert torch.equal(big(x), ref)              # slices of 38 / 37 on the two lanes
    assert torch.equal(big(x[:33]), ref[:33])
    assert torch.equal(big(x[:31]), ref[:31])    # below the threshold: caller's stream



def _bottleneck_operands(n, H, W, pitch, ldo, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(n, H, W, pitch, generator=g).abs() * 0.6).half().cuda()        # (a block's input went through a ReLU)
    w1 = torch.zeros(128, 256, dtype=torch.float16); w1[:64] = (torch.randn(64, 256, generator=g) * (2.0 / 256) ** 0.5).half()
    w2 = torch.zeros(128, 576, dtype=torch.float16); w2[:64] = (torch.randn(64, 576, generator=g) * (2.0 / 576) ** 0.5).half()
    w3 = (torch.randn(256, 64, generator=g) * 0.5 * (2.0 / 64) ** 0.5).half()
    b1 = torch.zeros(128); b1[:64] = torch.randn(64, generator=g) * 0.2
    b2 = torch.zeros(128); b2[:64] = torch.randn(64, generator=g) * 0.2
    b3 = torch.randn(256, generator=g) * 0.2
    return x, [t.cuda() for t in (w1, b1, w2, b2, w3, b3)]


def _bottleneck_three_kernels(x, ops, n, H, W, pitch, ldo):
    """conv1 -> conv2 -> conv3 + identity as the tower ran them through round 5: three launches, fp16 intermediates in HBM."""
    w1, b1, w2, b2, w3, b3 = ops
    L, st = _lib.lib(), _lib.stream_ptr()
    M = n * H * W
    t1 = torch.empty(M, 64, dtype=torch.float16, device="cuda")
    t2 = torch.empty(M, 64, dtype=torch.float16, device="cuda")
    out = torch.full((n, H, W, ldo), 7.0, dtype=torch.float16, device="cuda")
    assert L.lla_gemm_f16_ex(_lib.ptr(x), pitch, _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(t1), 64, None, 0, M, 128, 256,
                             _lib.LLA_EPI_RELU_F16, st) == 0
    assert L.lla_conv3x3_relu_f16(_lib.ptr(t1), n, H, W, 64, 64, _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(t2), 64, 128, st) == 0
    assert L.lla_gemm_f16_ex(_lib.ptr(t2), 64, _lib.ptr(w3), _lib.ptr(b3), _lib.ptr(out), ldo, _lib.ptr(x), pitch, M, 256, 64,
                             _lib.LLA_EPI_ADD_RELU_F16, st) == 0
    return out, t1, t2


@pytest.mark.parametrize("n,H,W,pitch,ldo", [(2, 56, 56, 256, 256), (1, 14, 14, 256, 256), (3, 28, 42, 320, 288), (70, 56, 56, 256, 256)])
def test_fused_layer1_bottleneck_against_the_three_kernels_and_float64(n, H, W, pitch, ldo):
    """`lla_rn50_bottleneck_f16` (csrc/bottleneck_fused.hip: conv1 -> conv2 -> conv3 + identity of a layer1 bottleneck in ONE
    kernel, 14 x 14 tiles with a recomputed halo, the 64-channel intermediates in LDS) against (a) the three kernels it replaces
    and (b) float64 with the intermediates rounded to fp16 where both paths round them.  The accumulation order differs from the
    GEMMs' (the accumulators start at the bias, other MFMA row order), so not bit-identical: within 2 fp16 ulps of the
    three-kernel bytes (ulp of max(|value|, 1): an intermediate that rounds the other way moves an output next to the ReLU's zero
    by more than its own ulp) and within the fp16 storage floor of float64 -- where both paths have the same error.  Image borders (conv2's zero padding is applied to t1, not to x), tiles that touch no
    border, more tiles than workgroups (the steady-state prefetch), pitches wider than the channels."""
    import torch.nn.functional as F
    x, ops = _bottleneck_operands(n, H, W, pitch, ldo, seed=n * 100 + H)
    w1, b1, w2, b2, w3, b3 = ops
    L = _lib.lib()
    out = torch.full((n, H, W, ldo), 7.0, dtype=torch.float16, device="cuda")
    rc = L.lla_rn50_bottleneck_f16(_lib.ptr(x), n, H, W, pitch, 256, _lib.ptr(w1), 256, _lib.ptr(b1), _lib.ptr(w2), 576,
                                   _lib.ptr(b2), _lib.ptr(w3), 64, _lib.ptr(b3), _lib.ptr(out), ldo, _lib.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    again = torch.full((n, H, W, ldo), 7.0, dtype=torch.float16, device="cuda")     # same launch, same bits (no atomics, fixed tile walk)
    assert L.lla_rn50_bottleneck_f16(_lib.ptr(x), n, H, W, pitch, 256, _lib.ptr(w1), 256, _lib.ptr(b1), _lib.ptr(w2), 576,
                                     _lib.ptr(b2), _lib.ptr(w3), 64, _lib.ptr(b3), _lib.ptr(again), ldo, _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, again)
    ref3, _, _ = _bottleneck_three_kernels(x, ops, n, H, W, pitch, ldo)
    torch.cuda.synchronize()
    if ldo > 256:
        assert bool((out[..., 256:] == 7.0).all())
    a, b = out[..., :256].float(), ref3[..., :256].float()
    ulp = torch.maximum(a.abs(), b.abs()).clamp_min(1.0) * 2.0 ** -10     # (outputs next to the ReLU's zero: absolute)
    d = (a - b).abs() / ulp
    assert float(d.max()) <= 2.0, float(d.max())
    assert float((d > 0).float().mean()) < 0.02            # the odd last-bit difference, nothing systematic
    # float64 on a sample of images, intermediates rounded to fp16 like both GPU paths
    idx = list(range(min(n, 2))) + ([n - 1] if n > 2 else [])
    xs = x[idx][..., :256].double().permute(0, 3, 1, 2)
    t1 = F.conv2d(xs, w1[:64].double().reshape(64, 256, 1, 1), b1[:64].double()).clamp_min(0).half().double()
    wk = w2[:64].double().reshape(64, 3, 3, 64).permute(0, 3, 1, 2)
    t2 = F.conv2d(t1, wk, b2[:64].double(), padding=1).clamp_min(0).half().double()
    ref = (F.conv2d(t2, w3.double().reshape(256, 64, 1, 1), b3.double()) + xs).clamp_min(0).permute(0, 2, 3, 1)
    err = (out[idx][..., :256].double() - ref).abs()
    tol = 1e-3 * ref.abs().clamp_min(1.0) + ref.abs() * 2.0 ** -11     # one fp16 rounding of the output + accumulation noise
    assert bool((err <= tol).all()), float((err / tol).max())
    # shapes it does not take are refused (the tower then runs the three kernels)
    args = lambda **kw: [kw.get("x", _lib.ptr(x)), n, kw.get("H", H), W, pitch, kw.get("cin", 256), _lib.ptr(w1), 256, _lib.ptr(b1),
                         _lib.ptr(w2), 576, _lib.ptr(b2), _lib.ptr(w3), 64, _lib.ptr(b3), kw.get("out", _lib.ptr(out)), ldo,
                         _lib.stream_ptr()]
    assert L.lla_rn50_bottleneck_f16(*args(H=H - 1)) != 0
    assert L.lla_rn50_bottleneck_f16(*args(cin=128)) != 0
    if pitch == ldo:
        assert L.lla_rn50_bottleneck_f16(*args(out=_lib.ptr(x))) != 0      # in place: halos would read written pixels


@pytest.mark.parametrize("n,H,W,pitch,ldo", [(2, 56, 56, 64, 256), (1, 14, 14, 64, 256), (3, 42, 28, 128, 320), (70, 56, 56, 64, 256)])
def test_fused_first_bottleneck_of_layer1_against_the_three_kernels_and_float64(n, H, W, pitch, ldo):
    """`lla_rn50_bottleneck_f16` with cin = 64: layer1's FIRST block -- conv1 (64 -> 64), conv2 (3x3), and conv3 + the downsample
    convolution as one 1x1 convolution over [t2 | x] (K = 128, no identity) -- against the three kernels the tower ran through
    round 5 (GEMM, direct 3x3, GEMM over the concatenated buffer) and float64 with fp16-rounded intermediates.  Tiles alternate
    between two LDS buffer sets (the input pixels stay resident until conv3 has multiplied them): odd and even tile counts,
    more tiles than workgroups."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(n * 7 + W)
    x = (torch.randn(n, H, W, pitch, generator=g).abs() * 0.6).half().cuda()
    w1 = torch.zeros(128, 64, dtype=torch.float16); w1[:64] = (torch.randn(64, 64, generator=g) * (2.0 / 64) ** 0.5).half()
    w2 = torch.zeros(128, 576, dtype=torch.float16); w2[:64] = (torch.randn(64, 576, generator=g) * (2.0 / 576) ** 0.5).half()
    wf = (torch.randn(256, 128, generator=g) * (1.0 / 128) ** 0.5).half()
    b1 = torch.zeros(128); b1[:64] = torch.randn(64, generator=g) * 0.2
    b2 = torch.zeros(128); b2[:64] = torch.randn(64, generator=g) * 0.2
    bf = torch.randn(256, generator=g) * 0.2
    w1, b1, w2, b2, wf, bf = (t.cuda() for t in (w1, b1, w2, b2, wf, bf))
    L, st = _lib.lib(), _lib.stream_ptr()
    out = torch.full((n, H, W, ldo), 7.0, dtype=torch.float16, device="cuda")
    assert L.lla_rn50_bottleneck_f16(_lib.ptr(x), n, H, W, pitch, 64, _lib.ptr(w1), 64, _lib.ptr(b1), _lib.ptr(w2), 576, _lib.ptr(b2),
                                     _lib.ptr(wf), 128, _lib.ptr(bf), _lib.ptr(out), ldo, st) == 0
    torch.cuda.synchronize()
    M = n * H * W
    t1 = torch.empty(M, 64, dtype=torch.float16, device="cuda")
    cat = torch.empty(M, 128, dtype=torch.float16, device="cuda")
    cat[:, 64:] = x.reshape(M, pitch)[:, :64]
    ref3 = torch.full((n, H, W, ldo), 7.0, dtype=torch.float16, device="cuda")
    assert L.lla_gemm_f16_ex(_lib.ptr(x), pitch, _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(t1), 64, None, 0, M, 128, 64, _lib.LLA_EPI_RELU_F16, st) == 0
    t2 = torch.empty(M, 64, dtype=torch.float16, device="cuda")
    assert L.lla_conv3x3_relu_f16(_lib.ptr(t1), n, H, W, 64, 64, _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(t2), 64, 128, st) == 0
    torch.cuda.synchronize()
    cat[:, :64] = t2
    assert L.lla_gemm_f16_ex(_lib.ptr(cat), 128, _lib.ptr(wf), _lib.ptr(bf), _lib.ptr(ref3), ldo, None, 0, M, 256, 128, _lib.LLA_EPI_RELU_F16, st) == 0
    torch.cuda.synchronize()
    if ldo > 256:
        assert bool((out[..., 256:] == 7.0).all())
    a, b = out[..., :256].float(), ref3[..., :256].float()
    d = (a - b).abs() / (torch.maximum(a.abs(), b.abs()).clamp_min(1.0) * 2.0 ** -10)
    assert float(d.max()) <= 2.0, float(d.max())
    assert float((d > 0).float().mean()) < 0.02
    idx = list(range(min(n, 2))) + ([n - 1] if n > 2 else [])
    xs = x[idx][..., :64].double().permute(0, 3, 1, 2)
    t1d = F.conv2d(xs, w1[:64].double().reshape(64, 64, 1, 1), b1[:64].double()).clamp_min(0).half().double()
    wk = w2[:64].double().reshape(64, 3, 3, 64).permute(0, 3, 1, 2)
    t2d = F.conv2d(t1d, wk, b2[:64].double(), padding=1).clamp_min(0).half().double()
    ref = F.conv2d(torch.cat([t2d, xs], 1), wf.double().reshape(256, 128, 1, 1), bf.double()).clamp_min(0).permute(0, 2, 3, 1)
    err = (out[idx][..., :256].double() - ref).abs()
    tol = 1e-3 * ref.abs().clamp_min(1.0) + ref.abs() * 2.0 ** -11
    assert bool((err <= tol).all()), float((err / tol).max())
