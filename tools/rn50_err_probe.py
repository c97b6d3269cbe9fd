"""RN50 tower error vs the fp32 oracle next to the fp16-storage emulation floor, a few seeds."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import rn50 as orn50                       # noqa: E402
from lossyless_amd.clip_rn50 import ModifiedResNet, synthetic_rn50_state_dict   # noqa: E402


def rel(z, ref):
    return np.linalg.norm(z - ref, axis=1) / np.linalg.norm(ref, axis=1)


torch.set_num_threads(32)
for wseed in (1, 2):
    sd = synthetic_rn50_state_dict(wseed)
    net = ModifiedResNet(sd, chunk=8).cuda()
    for seed in (3, 4):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(6, 224, 224, 3, generator=g).half()
        xc = x.permute(0, 3, 1, 2).float()
        ref = orn50.rn50_forward(sd, xc).numpy()
        emu = rel(orn50.rn50_forward(sd, xc, fp16_storage=True).numpy(), ref)
        hip = rel(net(x.cuda()).float().cpu().numpy(), ref)
        print(f"weights {wseed} images {seed}: hip max {hip.max():.2e} mean {hip.mean():.2e} | fp16-storage emulation max "
              f"{emu.max():.2e} mean {emu.mean():.2e}", flush=True)
