"""Disjoint parts of a GPU's CU mask for ranks that share the GPU.

STDLIB ONLY, and meant to be loaded BEFORE anything that may start the HIP / HSA runtime -- importing the package (torch,
torch.distributed, RCCL) is already too late on this stack: ``HSA_CU_MASK`` set after that is ignored (one GPU call of round 5
went into finding that out).  A launcher therefore loads this FILE, not the package::

    import importlib.util, os
    spec = importlib.util.spec_from_file_location("gpu_partition", os.path.join(repo, "lossyless_amd", "gpu_partition.py"))
    gp = importlib.util.module_from_spec(spec); spec.loader.exec_module(gp)
    gp.partition_shared_gpu(local_rank, local_world_size)

(`bench.py` does exactly this as the first thing in a rank.)
"""


def visible_gpus_without_hip():
    """GPUs this process will see, counted WITHOUT starting the HIP runtime (HSA_CU_MASK is read when it starts):
    the *_VISIBLE_DEVICES lists if set, else the KFD topology nodes that have SIMDs."""
    import glob
    import os
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip():
            return len([t for t in v.split(",") if t.strip()])
    n = 0
    for path in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")):
        try:
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
        except (OSError, ValueError):
            pass
    return n


def partition_shared_gpu(local_rank, local_world, xcds=8, cus_per_xcd=32, if_unknown=0):
    """Call BEFORE the first HIP call of the process.  When more ranks than GPUs are started on a node -- the dry run
    of the N > 1 path on a smaller box (``bench.py --gpus 2 --backend gloo`` on one MI355X, tests/test_gpu_configs.py)
    -- the ranks that share a GPU get disjoint CONTIGUOUS parts of its CU mask through ``HSA_CU_MASK``.  Why: two
    processes whose kernels run side by side on the same part of the chip lose kernel-boundary cache coherence now and
    then on this stack -- a record in 10^2 .. 10^7 images comes out one quantisation step off (24 of 24
    two-process runs differ without a mask and with interleaved mask bits, 0 of 82 with contiguous halves; which unit
    the halves separate was not established: ``HW_REG_XCC_ID`` shows all eight XCCs under every mask.  One process
    per GPU, the production configuration, is not affected).  Returns the mask it set, or None (one rank per GPU, a mask
    already in the environment, more than `xcds` sharers)."""
    import os
    n = visible_gpus_without_hip() or int(if_unknown)      # (`if_unknown`: GPUs to assume where neither source says)
    if n <= 0 or local_world <= n or os.environ.get("HSA_CU_MASK"):
        return None
    gpu = local_rank % n
    sharers = [r for r in range(local_world) if r % n == gpu]
    if len(sharers) > xcds:
        return None
    k = sharers.index(local_rank)
    lo, hi = k * xcds // len(sharers), (k + 1) * xcds // len(sharers)
    # (every CU named: this ROCr takes a comma list; a range `0-127` is ignored without a word -- profiles/r05_cu_mask_xcc_map.txt)
    mask = f"{gpu}:" + ",".join(str(i) for i in range(lo * cus_per_xcd, hi * cus_per_xcd))
    os.environ["HSA_CU_MASK"] = mask
    return mask
