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
