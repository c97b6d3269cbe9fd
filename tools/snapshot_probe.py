"""Debug probe (ablation build): find the first kernel whose output deviates when a two-lane tower pass is wrong.
The residual stream is snapshotted after ln_pre and after every residual GEMM; on a pass whose embeddings differ
from the baseline the snapshots are compared slot by slot.
    LLA_LIB=lossyless_amd/liblossyless_amd_ablation.so python tools/snapshot_probe.py [passes] [events]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hubconf
from lossyless_amd.compressor import SyntheticImages

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
want = int(sys.argv[2]) if len(sys.argv) > 2 else 4
comp, _ = hubconf.clip_compressor_b005(device="cuda", clip_weights="synthetic")
x = SyntheticImages(4096).device_batch(0, 1024, "cuda")
ROWS = 25600
snap = torch.zeros((2, 26, ROWS, 768), dtype=torch.float32, device="cuda")
PER = ROWS * (768 + 3072) * 2
snap16 = torch.zeros((2, 12, PER), dtype=torch.uint8, device="cuda")
if os.environ.get("SNAP16", "1") == "1":
    os.environ["LLA_VIT_SNAPSHOT16"] = str(snap16.data_ptr())
os.environ["LLA_VIT_SNAPSHOT"] = str(snap.data_ptr())
os.environ["LLA_VIT_SNAPSHOT_ROWS"] = str(ROWS)
z0 = comp.clip(x).clone()
torch.cuda.synchronize()
snap0 = snap.clone()
snap16_0 = snap16.clone()


def halves(t, lane, layer):
    h = t[lane, layer, :ROWS * 768 * 2].view(torch.float16).view(ROWS, 768)
    big = t[lane, layer, ROWS * 768 * 2:].view(torch.float16).view(ROWS, 3072)
    return h, big
for _ in range(3):
    z = comp.clip(x)
    torch.cuda.synchronize()
    assert torch.equal(z, z0) and torch.equal(snap, snap0), "baseline itself is not reproducible"
names = ["ln_pre"] + [f"L{l}.{k}" for l in range(12) for k in ("out_proj", "fc2")]
events = 0
for i in range(passes):
    z = comp.clip(x)
    torch.cuda.synchronize()
    if torch.equal(z, z0):
        continue
    events += 1
    rows = (z != z0).any(dim=1).nonzero().flatten().tolist()
    print(f"pass {i}: embedding rows {rows}", flush=True)
    for lane in range(2):
        for slot in range(25):
            d = (snap[lane, slot] != snap0[lane, slot])
            n = int(d.sum())
            if n:
                idx = d.nonzero()
                r = sorted(set(idx[:, 0].tolist())); c = sorted(set(idx[:, 1].tolist()))
                diff = (snap[lane, slot] - snap0[lane, slot])[d]
                print(f"  lane {lane} first deviating slot {slot} ({names[slot]}): {n} elements, {len(r)} rows {r[:12]}..{r[-1]} (image "
                      f"{lane * 512 + r[0] // 50}, token {r[0] % 50}), cols {c[:16]}{'...' if len(c) > 16 else ''} "
                      f"[{c[0]}..{c[-1]}], |diff| max {float(diff.abs().max()):.4g} min {float(diff.abs().min()):.4g}; "
                      f"values now {snap[lane, slot][d][:4].tolist()} baseline {snap0[lane, slot][d][:4].tolist()}", flush=True)
                if slot >= 2 and slot % 2 == 0:
                    layer = slot // 2 - 1
                    for nm, a, b in zip(("h after ln_2", "big after c_fc"), halves(snap16, lane, layer), halves(snap16_0, lane, layer)):
                        dd = a != b
                        if int(dd.sum()):
                            ii = dd.nonzero()
                            rr = sorted(set(ii[:, 0].tolist())); cc = sorted(set(ii[:, 1].tolist()))
                            df = (a.float() - b.float())[dd].abs()
                            print(f"    {nm}: {int(dd.sum())} elements differ, rows {rr[:8]}, cols [{cc[0]}..{cc[-1]}] ({len(cc)} distinct), "
                                  f"|diff| max {float(df.max()):.4g}; now {a[dd][:6].tolist()} baseline {b[dd][:6].tolist()}", flush=True)
                        else:
                            print(f"    {nm}: identical", flush=True)
                break
    if events >= want:
        break
print(f"{events} deviating passes in {i + 1}", flush=True)
