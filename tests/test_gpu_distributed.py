"""GPU: image-parallel `compress_dataset(distributed=True)` -- SURVEY.md 8(e) acceptance: the
file written by rank 0 of a 2-rank run equals the 1-rank file byte for byte.  Both ranks share
the single GPU of the test box and talk over gloo (the 8-GPU RCCL run is the driver's)."""
import hashlib
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
if int(sys.argv[3]) > 1:
    # both ranks on cuda:0: before ANYTHING that may start the HIP runtime is imported, every rank gets its own part of
    # the GPU's CU mask (lossyless_amd/gpu_partition.py, loaded as a file; DESIGN.md 5.4)
    import importlib.util
    spec = importlib.util.spec_from_file_location("gpu_partition", os.path.join(sys.argv[1], "lossyless_amd", "gpu_partition.py"))
    gp = importlib.util.module_from_spec(spec); spec.loader.exec_module(gp)
    lo = 128 * int(sys.argv[2])
    assert gp.partition_shared_gpu(int(sys.argv[2]), int(sys.argv[3]), if_unknown=1) == "0:" + ",".join(str(i) for i in range(lo, lo + 128))
import numpy as np, torch, torch.distributed as dist
import hubconf
from test_gpu_vit import synth_images
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
if world > 1:
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world)
comp, _ = hubconf.clip_compressor_b005(device="cuda:0", clip_weights="synthetic")

class DS(torch.utils.data.Dataset):
    def __init__(self):
        self.x = synth_images(37, seed=21).permute(0, 3, 1, 2).contiguous().float()
        self.y = (torch.arange(37) * 7) % 10
    def __len__(self): return 37
    def __getitem__(self, i): return self.x[i], self.y[i]

comp.compress_dataset(DS(), out + ".bin", label_file=out + ".npy",
                      kwargs_dataloader=dict(batch_size=8, num_workers=0), is_info=False,
                      distributed=world > 1)
if world > 1:
    dist.destroy_process_group()
print("RANK_DONE", rank)
"""


def _sha(p):
    with open(p, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def test_two_rank_file_equals_one_rank_file(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    one = str(tmp_path / "one")
    r = subprocess.run([sys.executable, str(script), ROOT, "0", "1", "0", one], capture_output=True,
                       text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    two = str(tmp_path / "two")
    port = str(29700 + os.getpid() % 200)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(k), "2", port, two],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for k in range(2)]
    outs = [p.communicate(timeout=280)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert _sha(one + ".bin") == _sha(two + ".bin")        # 37 images: shards of 19 + 18
    assert _sha(one + ".npy") == _sha(two + ".npy")
    import numpy as np
    assert np.load(two + ".npy").dtype == np.uint16


_RCCL_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from lossyless_amd import distributed as lla_dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=0, world_size=1,
                        device_id=torch.device("cuda:0"))
assert lla_dist.rank_world() == (0, 1) and lla_dist.shard_bounds(10, 0, 1) == (0, 10)
rng = np.random.default_rng(0)
body = rng.integers(0, 256, size=12345, dtype=np.uint8)
labels = rng.integers(0, 1000, size=77).astype(np.uint16)
b, l, n = lla_dist.gather_to_rank0(body, labels, 77, "cuda:0")      # device tensors through RCCL
assert n == 77 and np.array_equal(b, body) and np.array_equal(l, labels)
# records that never left the GPU (RecordStream.finish() of an on_device stream: what a sending rank hands over)
b, l, n = lla_dist.gather_to_rank0(torch.from_numpy(body).cuda(), labels, 77, "cuda:0")
assert n == 77 and np.array_equal(b, body) and np.array_equal(l, labels)
assert lla_dist.sends_from_device("cuda:0") is False      # rank 0 writes the file: its records go to the host
b, l, n = lla_dist.gather_to_rank0(np.zeros(0, np.uint8), np.zeros(0, np.uint16), 0, "cuda:0")
assert n == 0 and b.size == 0 and l.size == 0
t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
lla_dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_exchange_runs_on_the_rccl_backend(tmp_path):
    """The end-of-dataset exchange (int64 size all_gather + all_reduce, float64 max-reduce of
    the timing, barrier) on the real ``nccl`` = RCCL backend with device tensors.  One rank is
    all a 1-GPU box allows (RCCL refuses two ranks on one device); the multi-rank data flow is
    covered by the gloo tests, the collectives' device/dtype handling by this one."""
    script = tmp_path / "r.py"
    script.write_text(_RCCL_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), ROOT, str(29900 + os.getpid() % 90)], env=env,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


_NCCL2_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
if int(sys.argv[3]) > 1:
    # both ranks on cuda:0: before ANYTHING that may start the HIP runtime is imported, every rank gets its own part of
    # the GPU's CU mask (lossyless_amd/gpu_partition.py, loaded as a file; DESIGN.md 5.4)
    import importlib.util
    spec = importlib.util.spec_from_file_location("gpu_partition", os.path.join(sys.argv[1], "lossyless_amd", "gpu_partition.py"))
    gp = importlib.util.module_from_spec(spec); spec.loader.exec_module(gp)
    lo = 128 * int(sys.argv[2])
    assert gp.partition_shared_gpu(int(sys.argv[2]), int(sys.argv[3]), if_unknown=1) == "0:" + ",".join(str(i) for i in range(lo, lo + 128))
import numpy as np, torch, torch.distributed as dist
import hubconf
from lossyless_amd.compressor import SyntheticImages
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
dev = f"cuda:{rank}"
torch.cuda.set_device(rank)
if world > 1:
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:" + port, rank=rank, world_size=world,
                            device_id=torch.device(dev))
comp, _ = hubconf.clip_compressor_b005(device=dev, clip_weights="synthetic")
comp.compress_dataset(SyntheticImages(301, seed=4), out + ".bin", kwargs_dataloader=dict(batch_size=64),
                      is_info=False, distributed=world > 1)
if world > 1:
    dist.destroy_process_group()
print("RANK_DONE", rank)
"""


def _n_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.skipif(_n_gpus() < 2, reason="needs 2 GPUs: RCCL refuses two ranks on one device")
def test_two_rank_nccl_file_equals_one_rank_file(tmp_path):
    """SURVEY.md 8(e) on the real backend: two ranks on two GPUs, shards of 151 + 150 lazily
    generated images, records SENT to rank 0 over RCCL; file sha == 1-rank file sha."""
    script = tmp_path / "n.py"
    script.write_text(_NCCL2_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = str(tmp_path / "one")
    r = subprocess.run([sys.executable, str(script), ROOT, "0", "1", "0", one], env=env,
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    two = str(tmp_path / "two")
    port = str(29800 + os.getpid() % 90)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(k), "2", port, two], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for k in range(2)]
    outs = [p.communicate(timeout=280)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert _sha(one + ".bin") == _sha(two + ".bin")


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without a launcher must start two ranks itself and print ONE JSON
    line (the driver's contract).  RCCL when two GPUs are visible, else the gloo dry run of the
    same code path with both ranks on the one GPU; --dataset-images exercises the sharded
    compress_dataset + gather, and its bits/img must equal the 1-rank run's (same file)."""
    import json
    backend = "nccl" if _n_gpus() >= 2 else "gloo"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)

    def run(*args):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env,
                           capture_output=True, text=True, timeout=560)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        return json.loads(lines[0])

    common = ["--batch", "64", "--no-cpu-baseline", "--no-extra", "--backend", backend]
    two = run("--gpus", "2", "--steps", "3", "--warmup", "1", *common)
    assert two["n_gpus"] == 2 and two["metric"] == "encode_img_per_sec" and two["verified"] is True
    assert two["scaling"] == "weak" and two["value"] > 0
    # who ran where: one entry per rank with its own clock and its device's identity
    ranks = two["comm"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and all(r["img_per_sec"] > 0 and r["name"] for r in ranks)
    assert two["comm"]["distinct_devices"] == (2 if backend == "nccl" else 1)
    tr = two["timed_region"]
    assert tr["images"] == 2 * tr["images_per_gpu"] == 2 * 64 * tr["steps"] and tr["steps"] == 3 * tr["blocks"]
    d2 = run("--gpus", "2", "--dataset-images", "333", *common)
    d1 = run("--gpus", "1", "--dataset-images", "333", *common)
    assert d2["n_gpus"] == 2 and d1["n_gpus"] == 1
    assert d2["bits_per_img"] == d1["bits_per_img"] and d2["file_sha256"] == d1["file_sha256"]


def test_world_8_dry_run_of_the_timed_bench_path(tmp_path):
    """the TIMED path of bench.py (not --dataset-images) with eight ranks, as the driver will launch it on
    an 8-GPU node -- here all eight on the one GPU over gloo: one JSON line from rank 0, `comm.ranks` with eight entries
    (own clock, device identity, host CPUs and where they came from), and -- because the ranks share a GPU -- disjoint
    contiguous EIGHTHS of the CU mask, weak scaling arithmetic intact."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HSA_CU_MASK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--min-seconds", "0", "--batch", "256", "--no-cpu-baseline", "--no-extra"],
                       env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["verified"] is True and d["value"] > 0
    ranks = d["comm"]["ranks"]
    assert [x["rank"] for x in ranks] == list(range(8)) and d["comm"]["world_size"] == 8
    assert all(x["img_per_sec"] > 0 and x["host_cpus"] >= 1 and x["host_cpus_from"] in ("sysfs", "slice") for x in ranks)
    masks = [x["cu_mask"] for x in ranks]
    assert masks == ["0:%d..%d (32 CUs)" % (32 * k, 32 * k + 31) for k in range(8)], masks
    assert d["comm"]["distinct_devices"] == 1
    tr = d["timed_region"]
    assert tr["images"] == 8 * tr["images_per_gpu"]
