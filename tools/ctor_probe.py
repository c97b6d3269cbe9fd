import time, sys, os
sys.path.insert(0, os.getcwd())
t0 = time.perf_counter(); import torch; t1 = time.perf_counter()
import hubconf
t2 = time.perf_counter()
comp, _ = hubconf.clip_compressor_b005(device="cuda:0", clip_weights="synthetic")
torch.cuda.synchronize(); t3 = time.perf_counter()
from lossyless_amd.clip_vit import synthetic_vit_state_dict, pack_weights
t4 = time.perf_counter(); sd = synthetic_vit_state_dict(1); t5 = time.perf_counter(); blob = pack_weights(sd); t6 = time.perf_counter()
print(f"import torch {t1-t0:.1f}s | factory (synthetic weights) {t3-t2:.1f}s | of which: synth weights {t5-t4:.1f}s, pack_weights {t6-t5:.1f}s | torch threads {torch.get_num_threads()}", flush=True)
