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


def gather_bitstreams(blob, device=None, group=None, dst=0):
    """Gather one byte string per rank onto `dst`.  Returns the list of per-rank byte strings on `dst`
    (None elsewhere).  Sizes are exchanged with an all_gather, payloads with one padded gather."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([len(blob)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    payload = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if len(blob):
        payload[: len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    if rank == dst:
        recv = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.gather(payload, recv, dst=dst, group=group)
        return [bytes(recv[r][: sizes[r]].cpu().numpy().tobytes()) for r in range(world)]
    dist.gather(payload, None, dst=dst, group=group)
    return None


def concatenate_with_index(blobs):
    """Rank-0 side of the gather: one buffer + (offset, length) table, the shape a GLB BIN chunk wants."""
    offsets, pos = [], 0
    for b in blobs:
        offsets.append((pos, len(b)))
        pos += len(b)
    return b"".join(blobs), np.asarray(offsets, dtype=np.int64)
