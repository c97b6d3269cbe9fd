import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from lossyless_amd.clip_rn50 import ModifiedResNet, synthetic_rn50_state_dict
sd = synthetic_rn50_state_dict(1)
nets = [ModifiedResNet(sd, chunk=256).cuda() for _ in range(2)]
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(256, 224, 224, 3, generator=g, device="cuda").half()
def run(n):
    streams = [torch.cuda.Stream() for _ in range(n)]
    xs = x.chunk(n)
    def step():
        for s, net, xx in zip(streams, nets, xs):
            with torch.cuda.stream(s):
                net(xx)
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5
for n in (1, 2, 1, 2):
    dt = run(n)
    print(f"{n} stream(s): {dt*1e3:.2f} ms per 256 images = {256/dt:.0f} img/s")
