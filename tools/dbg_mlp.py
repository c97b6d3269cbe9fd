import sys, torch
sys.path.insert(0, '.')
from lossyless_amd.rates import HRateHyperprior
torch.manual_seed(11)
m = HRateHyperprior(512).eval(); m.update(force=True)
z = torch.randn(1024, 512, generator=torch.Generator().manual_seed(7)) * 2
def run(mod, zz):
    with torch.no_grad():
        z_in = mod.process_z_in(zz)
        side = mod.side_encoder(z_in)
        med = mod.entropy_bottleneck._medians()
        side_hat = torch.round(side - med) + med
        g = mod.z_encoder(side_hat)
        scales = g.chunk(2, -1)[0]
        idx = mod.gaussian_conditional.build_indexes(scales)
        return [t.cpu() for t in (z_in, side, side_hat, g, scales, idx)]
a = run(m, z)
mg = HRateHyperprior(512).eval(); mg.load_state_dict(m.state_dict()); mg = mg.cuda()
b = run(mg, z.cuda())
for name, x, y in zip(("z_in","side","side_hat","g","scales","idx"), a, b):
    x, y = x.float(), y.float()
    print(name, tuple(x.shape), tuple(y.shape), "cpu mean/std %.4g %.4g gpu %.4g %.4g maxdiff %.4g nan %d" % (x.mean(), x.std(), y.mean(), y.std(), (x-y).abs().max(), int(torch.isnan(y).sum())))
print("idx equal frac", float((a[5]==b[5]).float().mean()), a[5].flatten()[:8], b[5].flatten()[:8])
print("side_hat equal frac", float((a[2]==b[2]).float().mean()))
