"""The N>1 batch path on a GPU box: two ranks (gloo, both on cuda:0) code their LPT shares of a batch with one dmi_jobs_encode
each and gather the blobs onto rank 0 in mesh order — compared with the oracle there."""
import os
import socket
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _meshes():
    from draco_oxide_amd import synth
    return [synth.torus_mesh(10 + 3 * k, seed=900 + k, open_boundary=bool(k % 3 == 0)) for k in range(9)]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from draco_oxide_amd import distributed as dd
        blobs = dd.encode_meshes_sharded(_meshes())
        if rank == 0:
            q.put(blobs)
        else:
            assert blobs is None
    finally:
        dist.destroy_process_group()


def test_batch_sharded_over_two_ranks_matches_the_oracle():
    import draco_oxide_amd as dmi
    from draco_oxide_amd import distributed as dd
    from helpers import oracle_from_product_mesh
    meshes = _meshes()
    want = [oracle_from_product_mesh(m).encode() for m in meshes]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert got == want
    # world size 1 (no process group), in this process — after the children, so that nothing is spawned from a process that holds the GPU
    assert dmi.device_count() >= 1
    assert dd.encode_meshes_sharded(meshes) == want


def _nccl_single(port, q):
    import numpy as np
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        from draco_oxide_amd import distributed as dd
        blob = np.random.default_rng(5).integers(0, 256, size=100_003, dtype=np.uint8)
        a = dd.gather_bitstreams(blob.tobytes(), device=dev)
        b = dd.gather_bitstreams(blob, device=dev, as_bytes=False)
        b_ok = bool((b[0] == blob).all())   # (views of the receive buffer: valid until the next gather)
        c = dd.gather_blob_lists([blob.tobytes(), b"xyz"], [1, 0], 2, device=dev)
        q.put((a[0] == blob.tobytes(), b_ok, c == [b"xyz", blob.tobytes()]))
    finally:
        dist.destroy_process_group()


def test_gather_on_the_rccl_path_single_rank():
    """The device branch of the gather (RCCL group, device tensors, pinned host mirror) with a one-rank group — what a 1-GPU box can run."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_single, args=(_free_port(), q))
    p.start()
    got = q.get(timeout=300)
    p.join(timeout=300)
    assert p.exitcode == 0 and got == (True, True, True)


def test_single_process_multi_device_entries():
    """dmi_shard_meshes / dmi_meshes_prepare_devices / dmi_jobs_encode_devices: what a single-process caller (the Rust crate) uses to
    spread a batch over several GPUs.  On a 1-GPU box every mesh lands on device 0 (one group, plain dmi_jobs_encode underneath); with
    more devices the groups run concurrently.  Sections come back in mesh order and equal the oracle's."""
    import draco_oxide_amd as dmi
    from helpers import oracle_from_product_mesh
    meshes = _meshes()
    ndev = dmi.device_count()
    deal = dmi.shard_meshes(meshes, ndev)
    assert len(deal) == len(meshes) and set(deal) <= set(range(ndev))
    # LPT: loads differ by at most the heaviest mesh
    loads = [sum(len(m.faces) for m, d in zip(meshes, deal) if d == k) for k in range(ndev)]
    assert max(loads) - min(loads) <= max(len(m.faces) for m in meshes)
    jobs = dmi.meshes_prepare_devices(meshes, deal)
    outs = dmi.jobs_encode_devices(jobs)
    for m, j, o in zip(meshes, jobs, outs):
        assert j.header_and_connectivity + o == oracle_from_product_mesh(m).encode()
    for j in jobs:
        j.close()
    with pytest.raises(dmi.DracoMiError):
        dmi.meshes_prepare_devices(meshes, [ndev + 3] * len(meshes))   # a device that does not exist: an error code


def _transcode_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import test_distributed_cpu as tdc
        from draco_oxide_amd import gltf
        tm = {}
        res = gltf.transcode_files(tdc._fake_assets(), timings=tm)
        q.put((rank, tm["primitives_built"], None if res is None else [blobs for _, blobs in res], None if res is None else [bytes(g) for g, _ in res], tm.get("files_owned")))
    finally:
        dist.destroy_process_group()


def test_transcode_files_over_two_ranks_builds_only_its_share():
    """transcode_files in a two-rank job (round 6: FILES dealt by size before parsing, each rank runs dmi_transcode_assets over its own, finished files
    gathered): each rank builds and encodes only its files' primitives, rank 0 holds every file — byte for byte the files a single process writes."""
    import test_distributed_cpu as tdc
    from draco_oxide_amd import gltf
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_transcode_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    assert got[0][1] + got[1][1] == 5 and 0 < got[0][1] < 5 and got[1][2] is None
    single = gltf.transcode_files(tdc._fake_assets())
    assert got[0][2] == [blobs for _, blobs in single]
    assert got[0][3] == [bytes(g) for g, _ in single] and got[0][4] + got[1][4] == 3 and got[0][4] >= 1 and got[1][4] >= 1
