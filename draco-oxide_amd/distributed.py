"""Multi-GPU plumbing: one process per GPU, meshes sharded across ranks with no data-path collective;
the only exchange is the final gather of the finished bitstreams onto rank 0 (RCCL over xGMI when the
process group is `nccl`, gloo in the CPU tests).  Compressed output is ≈1–3 B per triangle, so the
gather is latency-bound, not bandwidth-bound."""
import numpy as np
import torch
import torch.distributed as dist


def shard_indices(num_items, rank, world_size, weights=None):
    """Greedy longest-processing-time partition of independent meshes over the ranks.
    Returns the item indices owned by `rank` (deterministic on every rank)."""
    if weights is None:
        return list(range(rank, num_items, world_size))
    order = sorted(range(num_items), key=lambda i: (-weights[i], i))
    load = [0] * world_size
    owner = [0] * num_items
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += weights[i]
    return [i for i in range(num_items) if owner[i] == rank]


_recv_cache = {}   # (world, cap, device) → (device receive buffer, pinned host mirror): reused across calls


def gather_bitstreams(blob, device=None, group=None, dst=0, as_bytes=True):
    """Gather one byte string per rank onto `dst`.  `blob`: bytes, or a uint8 numpy array / memoryview (no copy is made on the way
    to the device).  Returns the list of per-rank byte strings on `dst` (None elsewhere); with as_bytes=False the items are uint8
    numpy views of one host buffer that is reused by the next call (no per-rank copies: a pipeline that streams the blobs onwards).
    Sizes are exchanged with an all_gather, payloads with one padded gather into one contiguous buffer and one copy to the host."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    src = np.frombuffer(blob, dtype=np.uint8) if isinstance(blob, (bytes, bytearray, memoryview)) else np.ascontiguousarray(blob, dtype=np.uint8)
    n = torch.tensor([len(src)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = (max(max(sizes), 1) + 255) & ~255
    payload = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if len(src):
        with __import__("warnings").catch_warnings():
            __import__("warnings").simplefilter("ignore")   # (read-only buffer: it is only read)
            payload[: len(src)].copy_(torch.from_numpy(src), non_blocking=False)
    if rank != dst:
        dist.gather(payload, None, dst=dst, group=group)
        return None
    key = (world, cap, str(dev))
    if key not in _recv_cache:
        _recv_cache.clear()
        recv = torch.empty((world, cap), dtype=torch.uint8, device=dev)
        host = torch.empty((world, cap), dtype=torch.uint8, pin_memory=(dev.type == "cuda")) if dev.type == "cuda" else None
        _recv_cache[key] = (recv, host)
    recv, host = _recv_cache[key]
    dist.gather(payload, [recv[r] for r in range(world)], dst=dst, group=group)
    if host is not None:
        host.copy_(recv, non_blocking=False)
        arr = host.numpy()
    else:
        arr = recv.numpy()
    views = [arr[r, : sizes[r]] for r in range(world)]
    return [v.tobytes() for v in views] if as_bytes else views


def concatenate_with_index(blobs):
    """Rank-0 side of the gather: one buffer + (offset, length) table, the shape a GLB BIN chunk wants."""
    offsets, pos = [], 0
    for b in blobs:
        offsets.append((pos, len(b)))
        pos += len(b)
    return b"".join(blobs), np.asarray(offsets, dtype=np.int64)


def gather_blob_lists(local_blobs, local_indices, num_items, device=None, group=None, dst=0):
    """Batch form of the gather: every rank holds the finished bitstreams of the items it owns (`local_indices`, as dealt by
    shard_indices); `dst` receives all `num_items` of them in item order (None elsewhere).  One size exchange + one padded
    gather for the whole batch: each rank's payload is an index/length table followed by its bytes."""
    assert len(local_blobs) == len(local_indices)
    table = np.empty(1 + 2 * len(local_blobs), dtype=np.int64)
    table[0] = len(local_blobs)
    table[1::2] = np.asarray(local_indices, dtype=np.int64)
    table[2::2] = np.asarray([len(b) for b in local_blobs], dtype=np.int64)
    got = gather_bitstreams(table.tobytes() + b"".join(local_blobs), device=device, group=group, dst=dst)
    if got is None:
        return None
    out = [None] * num_items
    for payload in got:
        n = int(np.frombuffer(payload, dtype=np.int64, count=1)[0])
        t = np.frombuffer(payload, dtype=np.int64, count=1 + 2 * n)
        pos = (1 + 2 * n) * 8
        for k in range(n):
            idx, ln = int(t[1 + 2 * k]), int(t[2 + 2 * k])
            out[idx] = payload[pos:pos + ln]
            pos += ln
    missing = [i for i, b in enumerate(out) if b is None]
    if missing:
        raise RuntimeError(f"gather_blob_lists: items {missing[:8]} were owned by no rank")
    return out


def _rank_config(cfg, device):
    """The Config a rank encodes with: its HIP ordinal is the CUDA `device` it was given, else LOCAL_RANK (one process per GPU;
    with HIP_VISIBLE_DEVICES isolating one GPU per process the ordinal folds to 0).  A cfg that names another GPU than `device`
    is a caller bug: every rank would otherwise prepare and encode on device 0."""
    import copy
    import os
    from . import binding
    n_dev = max(binding.device_count(), 1)
    if device is not None and getattr(device, "type", "cpu") == "cuda" and device.index is not None:
        ordinal = int(device.index)
    else:
        ordinal = int(os.environ.get("LOCAL_RANK", "0")) % n_dev
    if cfg is None:
        return binding.Config(device=ordinal)
    if device is not None and getattr(device, "type", "cpu") == "cuda" and device.index is not None and cfg.device != ordinal:
        raise ValueError(f"encode_meshes_sharded: cfg.device={cfg.device} but the rank's device is cuda:{ordinal}")
    if device is None or getattr(device, "type", "cpu") != "cuda":
        cfg = copy.copy(cfg)
        cfg.device = cfg.device if cfg.device else ordinal   # an explicit non-zero ordinal is honoured
    return cfg


def encode_meshes_sharded(meshes, cfg=None, device=None, group=None, dst=0):
    """BASELINE configs[3] on N GPUs: the batch is dealt to the ranks by triangle count (shard_indices, LPT), every rank prepares
    and codes its share with ONE dmi_jobs_encode on its own GPU (no data-path collective), and the finished `.drc` blobs are
    gathered onto `dst` in mesh order (RCCL when the group is nccl).  Every rank passes the same `meshes` list (or at least the
    same triangle counts: a rank only touches the meshes it owns).  Returns the list of blobs on `dst`, None elsewhere."""
    from . import binding
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    cfg = _rank_config(cfg, device)
    mine = shard_indices(len(meshes), rank, world, weights=[len(m.faces) for m in meshes])
    blobs = []
    if mine:
        jobs = binding.meshes_prepare([meshes[i] for i in mine], cfg)
        sections = binding.jobs_encode(jobs)
        blobs = [j.header_and_connectivity + s for j, s in zip(jobs, sections)]
        for j in jobs:
            j.close()
    if world == 1:
        return blobs
    return gather_blob_lists(blobs, mine, len(meshes), device=device, group=group, dst=dst)
