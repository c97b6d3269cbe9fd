"""Precision stress of the tower against the fp32 oracle on weights with CLIP-like pathologies:
massive residual-stream outliers in a few channels, larger attention logits, non-trivial LayerNorm
affine.  Prints the per-image relative L2 error (bar: 1e-3).  GPU box only (needs oracle/)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lossyless_amd.clip_vit import VisionTransformer, synthetic_vit_state_dict  # noqa: E402
from oracle import vit as ovit  # noqa: E402
from test_gpu_vit import synth_images, _rel  # noqa: E402


def main():
    for name, outlier, qk_gain, w_gain in [("baseline", 0.0, 1.0, 1.0), ("outliers x30", 30.0, 1.0, 1.0),
                                           ("outliers x100", 100.0, 1.0, 1.0), ("sharp attention", 0.0, 4.0, 1.0),
                                           ("outliers x60 + sharp attention + 2x weights", 60.0, 3.0, 2.0)]:
        sd = synthetic_vit_state_dict(3)
        g = torch.Generator().manual_seed(9)
        for k in list(sd):
            if k.endswith("weight") and sd[k].dim() == 1:
                sd[k] = 1 + 0.3 * torch.randn(sd[k].shape, generator=g)
            elif k.endswith("bias"):
                sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
            elif sd[k].dim() == 2 and "in_proj" in k:
                sd[k] = sd[k] * qk_gain
            elif sd[k].dim() == 2:
                sd[k] = sd[k] * w_gain
        if outlier:
            sd["class_embedding"][[5, 300]] += outlier                 # massive cls-token channels
            sd["positional_embedding"][:, [77, 500]] += outlier * 0.5   # and two on every token
        x = synth_images(6, seed=12)
        ref = ovit.vit_b32_forward(sd, x.permute(0, 3, 1, 2).float()).numpy()
        z = VisionTransformer(sd).cuda()(x.cuda()).float().cpu().numpy()
        r = _rel(z, ref)
        em = _rel(ovit.vit_b32_forward(sd, x.permute(0, 3, 1, 2).float(), fp16_storage=True).numpy(), ref)
        print(f"{name:48s} HIP vs fp32: max {r.max():.2e} mean {r.mean():.2e} | fp16-storage emulation vs fp32: "
              f"max {em.max():.2e} mean {em.mean():.2e}")


if __name__ == "__main__":
    main()
