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


def kfd_gpu_nodes(sysfs="/sys"):
    """The KFD topology's GPU nodes (those with SIMDs) in node order -- the order HIP enumerates devices in -- as dicts of
    their integer properties (simd_count, simd_per_cu, num_xcc, drm_render_minor, ...).  Reads sysfs only: no HIP."""
    import glob
    import os
    nodes = []
    paths = glob.glob(os.path.join(sysfs, "class/kfd/kfd/topology/nodes/*/properties"))
    for path in sorted(paths, key=lambda q: int(os.path.basename(os.path.dirname(q)))):
        try:
            with open(path) as f:
                props = {}
                for line in f:
                    t = line.split()
                    if len(t) >= 2:
                        try:
                            props[t[0]] = int(t[1])
                        except ValueError:
                            pass
        except OSError:
            continue
        if props.get("simd_count", 0) > 0:
            props["node"] = int(os.path.basename(os.path.dirname(path)))
            nodes.append(props)
    return nodes


def visible_device_indices(n_nodes):
    """Indices into kfd_gpu_nodes() of the devices this process will see, from the *_VISIBLE_DEVICES lists (integer
    entries only; ROCR first, then HIP / CUDA on top of it, as the runtimes apply them); all of them if none is set."""
    import os
    idx = list(range(n_nodes))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is None or not v.strip():
            continue
        try:
            pick = [int(t) for t in v.split(",") if t.strip()]
        except ValueError:      # (UUID entries: cannot be resolved without the runtime)
            return None
        idx = [idx[i] for i in pick if 0 <= i < len(idx)]
        if var != "ROCR_VISIBLE_DEVICES":
            break
    return idx


def visible_gpus_without_hip(sysfs="/sys"):
    """GPUs this process will see, counted WITHOUT starting the HIP runtime (HSA_CU_MASK is read when it starts):
    the *_VISIBLE_DEVICES lists if set, else the KFD topology nodes that have SIMDs."""
    import os
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip():
            return len([t for t in v.split(",") if t.strip()])
    return len(kfd_gpu_nodes(sysfs))


def gpu_numa_cpus(device_index, sysfs="/sys"):
    """(numa node, sorted CPU ids) of the host socket HIP device `device_index` hangs off: KFD node -> drm_render_minor ->
    /sys/class/drm/renderD<minor>/device/numa_node -> /sys/devices/system/node/node<k>/cpulist.  (None, []) where any
    link of that chain is missing (no sysfs, numa_node == -1 on a single-socket host, UUID device lists)."""
    import os
    nodes = kfd_gpu_nodes(sysfs)
    vis = visible_device_indices(len(nodes))
    if not nodes or vis is None or not (0 <= device_index < len(vis)):
        return None, []
    minor = nodes[vis[device_index]].get("drm_render_minor", -1)
    try:
        with open(os.path.join(sysfs, f"class/drm/renderD{minor}/device/numa_node")) as f:
            numa = int(f.read().strip())
        if numa < 0:
            return None, []
        with open(os.path.join(sysfs, f"devices/system/node/node{numa}/cpulist")) as f:
            text = f.read().strip()
    except (OSError, ValueError):
        return None, []
    cpus = []
    for part in text.split(","):
        if "-" in part:
            lo, hi = part.split("-")
            cpus += range(int(lo), int(hi) + 1)
        elif part:
            cpus.append(int(part))
    return numa, sorted(cpus)


def partition_shared_gpu(local_rank, local_world, xcds=None, cus_per_xcd=None, if_unknown=0, sysfs="/sys"):
    """Call BEFORE the first HIP call of the process.  When more ranks than GPUs are started on a node -- the dry run
    of the N > 1 path on a smaller box (``bench.py --gpus 2 --backend gloo`` on one MI355X, tests/test_gpu_configs.py)
    -- the ranks that share a GPU get disjoint CONTIGUOUS parts of its CU mask through ``HSA_CU_MASK``.  Why: two
    processes whose kernels run side by side on the same part of the chip lose kernel-boundary cache coherence now and
    then on this stack -- a record in 10^2 .. 10^7 images comes out one quantisation step off (24 of 24
    two-process runs differ without a mask and with interleaved mask bits, 0 of 82 with contiguous halves; which unit
    the halves separate was not established: ``HW_REG_XCC_ID`` shows all eight XCCs under every mask.  One process
    per GPU, the production configuration, is not affected).  The mask geometry (XCDs x CUs per XCD) is read from the
    KFD topology of the GPU in question (num_xcc, simd_count / simd_per_cu) and falls back to the MI355X's 8 x 32 where
    that cannot be read; rank r is assumed to use GPU r % n (what bench.py and the tests do).  Returns the mask it
    set, or None (one rank per GPU, a mask already in the environment, more sharers than XCDs)."""
    import os
    n = visible_gpus_without_hip(sysfs) or int(if_unknown)      # (`if_unknown`: GPUs to assume where neither source says)
    if n <= 0 or local_world <= n or os.environ.get("HSA_CU_MASK"):
        return None
    gpu = local_rank % n
    if xcds is None or cus_per_xcd is None:
        nodes = kfd_gpu_nodes(sysfs)
        vis = visible_device_indices(len(nodes))
        props = nodes[vis[gpu]] if nodes and vis is not None and gpu < len(vis) else {}
        x = props.get("num_xcc", 0) or 8
        cus = props.get("simd_count", 0) // max(props.get("simd_per_cu", 4), 1) or 256
        xcds = xcds or x
        cus_per_xcd = cus_per_xcd or max(cus // xcds, 1)
    sharers = [r for r in range(local_world) if r % n == gpu]
    if len(sharers) > xcds:
        return None
    k = sharers.index(local_rank)
    lo, hi = k * xcds // len(sharers), (k + 1) * xcds // len(sharers)
    # (every CU named: this ROCr takes a comma list; a range `0-127` is ignored without a word -- profiles/r05_cu_mask_xcc_map.txt)
    mask = f"{gpu}:" + ",".join(str(i) for i in range(lo * cus_per_xcd, hi * cus_per_xcd))
    os.environ["HSA_CU_MASK"] = mask
    return mask
