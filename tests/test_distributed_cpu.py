"""world_size-2 gloo tests of the N>1 path: shard assignment and the bitstream gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import draco_oxide_amd  # noqa: F401  (registers the package)
from draco_oxide_amd import distributed as dd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(100 + rank)
        blob = rng.integers(0, 256, size=1000 + 777 * rank, dtype=np.uint8).tobytes()
        got = dd.gather_bitstreams(blob)
        empty = dd.gather_bitstreams(b"" if rank == 1 else b"x")
        # batch form: 7 items dealt by weight, each rank contributes the blobs of the items it owns
        weights = [50, 900, 20, 400, 400, 10, 70]
        mine = dd.shard_indices(len(weights), rank, world, weights=weights)
        lists = dd.gather_blob_lists([bytes([i]) * weights[i] for i in mine], mine, len(weights))
        if rank == 0:
            q.put((got, empty, lists))
        else:
            assert got is None and empty is None and lists is None
    finally:
        dist.destroy_process_group()


def test_gather_bitstreams_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, empty, lists = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(2):
        want = np.random.default_rng(100 + r).integers(0, 256, size=1000 + 777 * r, dtype=np.uint8).tobytes()
        assert got[r] == want
    assert empty == [b"x", b""]
    assert lists == [bytes([i]) * w for i, w in enumerate([50, 900, 20, 400, 400, 10, 70])]
    buf, index = dd.concatenate_with_index(got)
    assert len(buf) == sum(len(b) for b in got) and index.tolist() == [[0, 1000], [1000, 1777]]


def test_shard_indices_cover_and_balance():
    weights = [200_000, 5_000, 120_000, 90_000, 3_000, 150_000, 60_000, 2_000, 40_000, 175_000]
    for world in (1, 2, 4, 8):
        owned = [dd.shard_indices(len(weights), r, world, weights) for r in range(world)]
        flat = sorted(i for o in owned for i in o)
        assert flat == list(range(len(weights)))
        loads = [sum(weights[i] for i in o) for o in owned]
        assert max(loads) <= sum(weights) / world + max(weights)
    assert dd.shard_indices(5, 1, 2) == [1, 3]


def test_shard_meshes_matches_the_python_deal_and_rank_config(monkeypatch):
    """dmi_shard_meshes (C, for single-process callers) deals exactly like distributed.shard_indices (LPT by triangle count); and
    encode_meshes_sharded picks each rank's HIP ordinal from LOCAL_RANK / its torch device (every rank used device 0)."""
    import numpy as np
    import pytest
    import torch
    import draco_oxide_amd as dmi
    from draco_oxide_amd import binding, distributed as dd
    rng = np.random.default_rng(4)
    meshes = []
    for k in range(23):
        f = int(rng.integers(1, 400))
        meshes.append(dmi.Mesh(np.zeros((f, 3), np.uint32), [dmi.Attribute(np.zeros((1, 3), np.float32), dmi.ATT_POSITION)]))
    for world in (1, 2, 3, 8):
        deal = dmi.shard_meshes(meshes, world)
        for r in range(world):
            assert [i for i, d in enumerate(deal) if d == r] == dd.shard_indices(len(meshes), r, world, weights=[len(m.faces) for m in meshes])
    monkeypatch.setattr(binding, "device_count", lambda: 8)
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert dd._rank_config(None, None).device == 5
    assert dd._rank_config(None, torch.device("cuda", 3)).device == 3
    assert dd._rank_config(dmi.Config(device=3, pos_bits=14), torch.device("cuda", 3)).pos_bits == 14
    with pytest.raises(ValueError):
        dd._rank_config(dmi.Config(device=1), torch.device("cuda", 3))
    monkeypatch.setenv("LOCAL_RANK", "9")
    assert dd._rank_config(None, torch.device("cpu")).device == 1   # 9 % 8 visible devices
