import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from lossyless_amd.clip_vit import VisionTransformer, synthetic_vit_state_dict
net = VisionTransformer(synthetic_vit_state_dict(1)).cuda()
g = torch.Generator(device="cuda").manual_seed(0)
for B in [int(a) for a in sys.argv[1:]]:
    X = torch.randn(B, 224, 224, 3, generator=g, device="cuda").half()
    for _ in range(3): z = net(X)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = max(5, 20480 // B)
    for _ in range(n): z = net(X)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    w = torch.arange(z.numel(), device="cuda") % 8191 + 1
    print(f"B={B}: {dt*1e3:8.3f} ms  {B/dt:9.0f} img/s  checksum {int((z.view(torch.int16).long().flatten()*w).sum())}")
