import os, sys, time, json
sys.path.insert(0, os.getcwd())
import torch, hubconf, bench
comp, _ = hubconf.clip_compressor_b005(device="cuda:0", clip_weights="synthetic")
for n in (8192, 32768):
    print(n, json.dumps(bench.stl10_shaped_leg(comp, "cuda:0", n=n)))
