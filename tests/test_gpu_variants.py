"""GPU: every alternative GEMM code path (the A/B switches of the -DLLA_ABLATION build: `make -C lossyless_amd/csrc
ablation`, loaded through LLA_LIB) must give the same answers, bit for bit, as the PRODUCT library's one path --
"default" below runs on the product library, every other variant on the ablation build with its switches set.  The
switches are read once per process, so each variant runs in its own interpreter.  The product library itself reads no
environment variable and holds no kernel that is off by default (last test)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, ablation_env

pytestmark = pytest.mark.gpu

_SCRIPT = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch
from lossyless_amd import _lib
from lossyless_amd.clip_vit import VisionTransformer, synthetic_vit_state_dict
from test_gpu_vit import synth_images, _rel
L = _lib.lib()
g = torch.Generator().manual_seed(1)
sums = []
# the last four are large enough (M >= 9000) for the persistent kernel: 256- and 320-row tiles,
# ragged last row tile (direct epilogue) next to full ones (LDS-staged epilogue)
# ... and four whose M is a whole number of 256-row tiles: the four-wave kernel (gemm_q4.hip) takes those
for (M, N, K, epi) in [(1000, 768, 3072, 2), (777, 2304, 768, 0), (512, 3072, 768, 1), (100, 512, 768, 0),
                       (17001, 2304, 768, 0), (12837, 768, 768, 2), (9100, 3072, 768, 1), (17000, 768, 3072, 2),
                       (17408, 2304, 768, 0), (12800, 768, 768, 2), (9216, 3072, 768, 1), (17408, 768, 3072, 2),
                       # ... and two with fewer tiles than CUs: one tile per workgroup (the pipelined epilogue's first / last
                       # tile paths: no pending fragment on entry, fragment 3 stored after the loop)
                       (9216, 256, 768, 0), (9216, 512, 1024, 1)]:
    A = (torch.randn(M, K, generator=g) * 0.5).half().cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).half().cuda()
    bias = torch.randn(N, generator=g).cuda()
    ref = A.double() @ W.double().t() + bias.double()
    if epi == 1:
        ref = ref * torch.sigmoid(1.702 * ref)
    if epi == 2:
        C = torch.randn(M, N, generator=g).cuda(); ref = ref + C.double()
    else:
        C = torch.empty(M, N, dtype=torch.float16, device="cuda")
    rc = L.lla_gemm_f16(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(C), M, N, K, epi, _lib.stream_ptr())
    assert rc == 0
    err = (C.double() - ref).abs()
    assert bool((err <= ref.abs() * 2 ** -10 + 2e-3).all()), (M, N, K, epi, float(err.max()))
    bits = C.view(torch.int16 if C.dtype == torch.float16 else torch.int32).long().flatten()
    w = torch.arange(bits.numel(), device="cuda") % 8191 + 1
    sums += [int(bits.sum()), int((bits * w).sum())]
want = np.load(os.path.join(sys.argv[1], "tests", "golden", "vit_synth_z.npy"))
z = VisionTransformer(synthetic_vit_state_dict(1)).cuda()(synth_images(4).cuda()).float().cpu().numpy()
assert _rel(z, want).max() < 1e-3
np.savez(sys.argv[2], z=z, sums=np.array(sums, dtype=np.int64))
print("VARIANT_OK")
"""

VARIANTS = {
    "default": {},
    "ablation_build_defaults": {"LLA_NOTHING": "0"},
    "tile128_regstage": {"LLA_GEMM_TILE": "128", "LLA_GEMM_GLDS": "0", "LLA_GEMM_Q4": "0", "LLA_GEMM_W8": "0"},
    "tile128_glds": {"LLA_GEMM_TILE": "128", "LLA_GEMM_Q4": "0", "LLA_GEMM_W8": "0"},
    "tile256_asm": {"LLA_GEMM_TILE": "256", "LLA_GEMM_Q4": "0", "LLA_GEMM_W8": "0"},
    "ping_pong_only": {"LLA_GEMM_Q4": "0", "LLA_GEMM_W8": "0"},
    "every_kernel_top_down": {"LLA_VIT_ZIGZAG": "0"},
    "four_wave_serial_epilogue": {"LLA_Q4_PIPE": "0", "LLA_GEMM_W8": "0"},
    "four_wave_dma_schedule_0": {"LLA_Q4_SCHED": "0", "LLA_GEMM_W8": "0"},
    "four_wave_dma_schedule_2": {"LLA_Q4_SCHED": "2", "LLA_GEMM_W8": "0"},
    "lockstep_persistent": {"LLA_GEMM_PP": "0", "LLA_GEMM_Q4": "0", "LLA_GEMM_W8": "0"},
    "persistent_kb32": {"LLA_GEMM_PP": "0", "LLA_GEMM_KB": "32", "LLA_GEMM_Q4": "0", "LLA_GEMM_W8": "0"},
    "one_tile_per_block": {"LLA_GEMM_PP": "0", "LLA_GEMM_PERSIST": "0", "LLA_GEMM_Q4": "0", "LLA_GEMM_W8": "0"},
    "direct_epilogue": {"LLA_GEMM_PP": "0", "LLA_GEMM_EPILOGUE": "direct", "LLA_GEMM_Q4": "0", "LLA_GEMM_W8": "0"},
    "pp_staged_fp16_epilogue": {"LLA_GEMM_EPILOGUE": "staged", "LLA_GEMM_Q4": "0", "LLA_GEMM_W8": "0"},
    "no_tall_tiles": {"LLA_GEMM_TALL": "0", "LLA_GEMM_Q4": "0", "LLA_GEMM_W8": "0"},
    "full_persistent_grid": {"LLA_GEMM_BALANCED": "0", "LLA_GEMM_Q4": "0", "LLA_GEMM_W8": "0"},
    "no_last_block_pruning": {"LLA_VIT_PRUNE_LAST": "0"},
    "small_chunks": {"LLA_VIT_CHUNK": "3"},
    # round 6: the eight-wave kernel on v_mfma_f32_16x16x32_f16 (gemm_w8.hip) takes the large fp16-output GEMMs in the
    # product; the variants above that pin an older path switch it off (LLA_GEMM_W8=0), these exercise it: off alone
    # (= the round-5 selection: the four-wave kernel's pipelined fp16 epilogues), its pipelined epilogue, and at every M
    # (single ragged tiles, M < 256) -- the small MFMA shape gives the bits of the large one
    "eight_wave_off": {"LLA_GEMM_W8": "0"},
    "eight_wave_pipelined_epilogue": {"LLA_W8_PIPE": "1"},
    "eight_wave_at_every_m": {"LLA_GEMM_W8": "2"},
}


# kernels retired from the product in rounds 2-4 that live on in the tools/ build only (round-1 tiles, the lock-step and
# one-tile-per-workgroup persistent kernels, their epilogue / grid options): `-m "gpu and slow"` since round 6 -- six
# seconds of interpreter start each, and the driver's GPU step has a time limit (VERDICT r5 #4b)
_RETIRED = {"tile128_regstage", "tile128_glds", "tile256_asm", "lockstep_persistent", "persistent_kb32", "one_tile_per_block",
            "direct_epilogue", "no_tall_tiles", "full_persistent_grid", "four_wave_dma_schedule_0", "four_wave_dma_schedule_2",
            "pp_staged_fp16_epilogue", "every_kernel_top_down", "small_chunks"}


@pytest.mark.parametrize("name", [pytest.param(n, marks=pytest.mark.slow) if n in _RETIRED else n for n in VARIANTS])
def test_gemm_variant_matches(name, tmp_path):
    script = tmp_path / "v.py"
    script.write_text(_SCRIPT)
    env = dict(os.environ) if name == "default" else ablation_env(**VARIANTS[name])
    env.pop("LLA_LIB", None) if name == "default" else None
    out = tmp_path / "z.npz"
    r = subprocess.run([sys.executable, str(script), ROOT, str(out)], env=env, capture_output=True,
                       text=True, timeout=280)
    assert r.returncode == 0 and "VARIANT_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    # every path accumulates K in the same order -> embeddings and raw GEMM outputs (compared
    # through two checksums of their bit patterns) are bit-identical across variants
    import numpy as np
    ref = tmp_path.parent / "variant_default.npz"
    got = np.load(out)
    if name == "default":
        np.savez(ref, z=got["z"], sums=got["sums"])
    elif ref.exists():
        want = np.load(ref)
        assert np.array_equal(got["z"], want["z"]), f"{name} differs bitwise from the default path"
        assert np.array_equal(got["sums"], want["sums"]), f"{name}: GEMM outputs differ bitwise from the default path"


def test_the_product_library_reads_no_environment_and_holds_no_optional_kernel(tmp_path):
    """every switch of rounds 1-4 set at once changes nothing in the product library (same bits as the
    default run), the library does not import getenv at all, and the kernels that are off by default -- the retired
    ones (two-workgroups-per-CU / first four-wave GEMM, algebraic LayerNorm fusion), the probe instantiations and the
    alternative instantiations the variants above select -- are not compiled into it."""
    import numpy as np
    from lossyless_amd import _lib
    script = tmp_path / "v.py"
    script.write_text(_SCRIPT)
    env = dict(os.environ, LLA_VIT_STREAMS="2", LLA_GEMM_DUO="1", LLA_GEMM_QUAD="1", LLA_VIT_LN_FUSE="1", LLA_Q4_DBG="13",
               LLA_GEMM_Q4="0", LLA_GEMM_TILE="128", LLA_GEMM_PP="0", LLA_Q4_PIPE="0", LLA_Q4_SCHED="2", LLA_VIT_ZIGZAG="0",
               LLA_VIT_CHUNK="3", LLA_GEMM_EPILOGUE="direct", LLA_VIT_PRUNE_LAST="0", LLA_GEMM_KB="32", LLA_GEMM_W8="0",
               LLA_W8_PIPE="1", LLA_W8_DBG="13")
    env.pop("LLA_LIB", None)
    out = tmp_path / "z.npz"
    r = subprocess.run([sys.executable, str(script), ROOT, str(out)], env=env, capture_output=True, text=True,
                       timeout=280)
    assert r.returncode == 0 and "VARIANT_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    ref = tmp_path.parent / "variant_default.npz"
    if ref.exists():
        want, got = np.load(ref), np.load(out)
        assert np.array_equal(got["z"], want["z"]) and np.array_equal(got["sums"], want["sums"])
    # the product's translation units -- and every header they include -- carry no A/B switch site and no
    # conditional region for the tools/ builds: those live under csrc/ablation/ (compiled by `make ablation` / `make probes`
    # INSTEAD of the product's launchers and switch values)
    import re
    csrc = os.path.join(ROOT, "lossyless_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    listed = lambda var: re.search(rf"^{var}\s*=\s*(.*)$", mk, re.M).group(1).split()
    product = listed("SHARED") + listed("PROD_ONLY")
    assert "tower.hip" in product and "gemm_pp.hip" in product and not any(f.startswith("ablation/") for f in product)
    headers = [f for f in os.listdir(csrc) if f.endswith(".h")]
    for fn in product + headers:
        src = open(os.path.join(csrc, fn)).read()
        code = re.sub(r"//.*", "", src)
        assert not re.search(r"LLA_ABLATION|LLA_PROBES|getenv", code), f"{fn}: a tools/-build conditional or an environment read"
        assert '#include "ablation/' not in src, fn
    assert not os.path.exists(os.path.join(csrc, "vit.hip")), "the round-1..5 monolith is back"
    undefined = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in undefined, "the product library imports getenv"
    names = subprocess.run(["strings", "-a", _lib.LIB_PATH], capture_output=True, text=True).stdout
    for gone in ("gemm_duo_kernel", "gemm_quad_kernel", "ln_stats_kernel", "shadow_compare_kernel"):
        assert gone not in names, f"{gone} is still compiled into the product library"
    kernels = sorted({l for l in names.splitlines() if l.startswith("_ZN3lla") and "kernel" in l and "device_stub" in l})
    q4 = [k for k in kernels if "gemm_q4_kernel" in k]
    assert len(q4) == 2, q4            # residual, residual + LayerNorm (the fp16-output layers run on the eight-wave kernel)
    w8 = [k for k in kernels if "gemm_w8_kernel" in k]
    assert len(w8) == 2, w8            # fp16 and QuickGELU, serial epilogue: one each
    assert not any("gemm_f16_kernel" in k and "Lb0E" in k for k in kernels)        # register-staged 128 x 128 tiles
    assert not any("gemm_persistent_kernel" in k and "Li32E" in k for k in kernels)   # K-tiles of 32
