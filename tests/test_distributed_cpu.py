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


# ---- the transcode driver shards BEFORE it builds (io/gltf/transcoder.rs:134-151 around encode.rs:932-955) ----
def _fake_assets():
    """Three small GLBs (5 primitives) made with the test module's asset writer."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gltf as tg
    from draco_oxide_amd import gltf
    files = [[tg._prim(9, 1), tg._prim(14, 2, normals=False, index_type="u16")], [tg._prim(20, 4, feat=True)], [tg._prim(12, 7), tg._prim(6, 8, uvs=False)]]
    return [gltf.write_glb(*tg._make_asset(prims, interleave=bool(i % 2))) for i, prims in enumerate(files)]


def _fake_encode_raw_batch(raws, cfg=None, pipeline=True, timings=None, weights=None, on_done=None, keep=None):
    """Stands in for the device: a "blob" that names the primitive by a digest of the bytes its views reference."""
    import hashlib
    out = []
    for r in ([raws(i) for i in range(len(weights))] if callable(raws) else raws):
        h = hashlib.sha256()
        for rows, t, d, par in r.atts:
            h.update(np.ascontiguousarray(rows).tobytes())
        h.update(np.ascontiguousarray(r.indices).tobytes())
        out.append((h.digest(), len(r.indices) // 3, r.atts[0][0].shape[0]))
    if on_done:
        on_done(list(range(len(out))), out)
    return out


def _transcode_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from draco_oxide_amd import gltf
        built = []
        real = gltf.primitive_to_raw
        gltf.primitive_to_raw = lambda doc, binary, prim: (built.append(gltf.primitive_weight(doc, prim)), real(doc, binary, prim))[1]
        gltf.encode_raw_batch = _fake_encode_raw_batch
        tm = {}
        res = gltf.transcode_files(_fake_assets(), timings=tm)
        q.put((rank, built, tm["primitives_built"], res))
    finally:
        dist.destroy_process_group()


def test_transcode_files_shards_before_building_gloo_world2():
    from draco_oxide_amd import gltf
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_transcode_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, built0, n0, res0), (_, built1, n1, res1) = got
    # every rank touched only the primitives shard_indices gave it: the two shares partition the five primitives by the JSON's counts
    weights = [2 * 9 * 9, 2 * 14 * 14, 2 * 20 * 20, 2 * 12 * 12, 2 * 6 * 6]
    want0 = [weights[i] for i in dd.shard_indices(5, 0, 2, weights=weights)]
    want1 = [weights[i] for i in dd.shard_indices(5, 1, 2, weights=weights)]
    assert built0 == want0 and built1 == want1 and n0 == len(want0) and n1 == len(want1) and n0 + n1 == 5
    assert res1 is None and len(res0) == 3
    # rank 0 assembled every file around the gathered blobs: the same GLBs as a single process makes with the same stand-in encoder
    gltf_encode = gltf.encode_raw_batch
    gltf.encode_raw_batch = _fake_encode_raw_batch
    was = os.environ.get("DMI_TRANSCODE_PYTHON")
    os.environ["DMI_TRANSCODE_PYTHON"] = "1"      # (the stand-in replaces the Python stage loop; a single process otherwise runs the library's own: dmi_transcoder)
    try:
        single = gltf.transcode_files(_fake_assets())
    finally:
        gltf.encode_raw_batch = gltf_encode
        if was is None:
            del os.environ["DMI_TRANSCODE_PYTHON"]
        else:
            os.environ["DMI_TRANSCODE_PYTHON"] = was
    assert [g for g, _ in res0] == [g for g, _ in single] and [b for _, b in res0] == [b for _, b in single]
    doc, _ = gltf.read_glb(res0[0][0])
    assert doc["accessors"][doc["meshes"][0]["primitives"][1]["indices"]]["count"] == 2 * 14 * 14 * 3


# ---- round 6: the rank-sharded transcode deals FILES by size before anything is parsed; finished files travel ----
def _fake_transcode_assets(assets, cfg=None, devices=None):
    """Stands in for dmi_transcode_assets: the "GLB" of a file is a digest of its bytes + two "blobs" inside it (memoryviews of the file, like the library's)."""
    import hashlib
    out = []
    for a in assets:
        h = hashlib.sha256(bytes(a)).digest()
        glb = memoryview(b"glTF" + h * 3)
        out.append((glb, [glb[4:36], glb[40:50]]))
    return out, {"parse_ms": 1.0, "build_ms": 2.0, "prepare_ms": 3.0, "encode_ms": 4.0, "assemble_ms": 5.0, "primitives": len(assets)}


def _files_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from draco_oxide_amd import gltf
        seen = []
        gltf._TRANSCODE_ASSETS = lambda assets, cfg=None, devices=None: (seen.extend(len(a) for a in assets), _fake_transcode_assets(assets, cfg, devices))[1]
        gltf.load_document = lambda src: (_ for _ in ()).throw(AssertionError("a rank parsed a document"))   # nothing is parsed outside the library's own loop
        rng = np.random.default_rng(3)
        files = [bytes(rng.integers(0, 255, size=int(n), dtype=np.uint8)) for n in (5000, 100, 70000, 3000, 42000, 900, 15000)]
        tm = {}
        res = gltf.transcode_files(files, timings=tm)
        q.put((rank, seen, tm.get("files_owned"), res))
    finally:
        dist.destroy_process_group()


def test_transcode_files_deals_files_by_size_before_parsing_gloo_world2():
    import hashlib
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_files_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, seen0, n0, res0), (_, seen1, n1, res1) = got
    sizes = [5000, 100, 70000, 3000, 42000, 900, 15000]
    own0, own1 = dd.shard_indices(7, 0, 2, weights=sizes), dd.shard_indices(7, 1, 2, weights=sizes)
    assert seen0 == [sizes[i] for i in own0] and seen1 == [sizes[i] for i in own1] and n0 == len(own0) and n1 == len(own1)     # each rank's library call saw ITS files only
    assert abs(sum(seen0) - sum(seen1)) <= max(sizes)                                                                   # LPT by bytes
    assert res1 is None and len(res0) == 7
    rng = np.random.default_rng(3)
    files = [bytes(rng.integers(0, 255, size=int(n), dtype=np.uint8)) for n in sizes]
    for f, (glb, blobs) in zip(files, res0):                                                                             # finished files in input order, blobs cut out of them
        h = hashlib.sha256(f).digest()
        assert glb == b"glTF" + h * 3 and blobs == [glb[4:36], glb[40:50]]


def _manifest_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from draco_oxide_amd import gltf
        gltf._TRANSCODE_ASSETS = _fake_transcode_assets
        rng = np.random.default_rng(4)
        files = [bytes(rng.integers(0, 255, size=int(n), dtype=np.uint8)) for n in (5000, 100, 70000, 3000, 42000, 900, 15000)]
        tm = {}
        res = gltf.transcode_files(files, timings=tm, gather="manifest", copy=True)
        q.put((rank, res, tm["manifest"][0].tolist(), tm["manifest"][1].tolist(), tm.get("files_owned")))
    finally:
        dist.destroy_process_group()


def test_transcode_files_manifest_mode_keeps_files_on_their_ranks_gloo_world2():
    """gather="manifest" (round 6): a rank's finished files stay with it — a transcoder's outputs are files, each rank writes its own — and sizes + xxh64 digests of ALL
    files reach every rank in one all_reduce."""
    import hashlib
    import xxhash
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_manifest_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sizes = [5000, 100, 70000, 3000, 42000, 900, 15000]
    rng = np.random.default_rng(4)
    files = [bytes(rng.integers(0, 255, size=int(n), dtype=np.uint8)) for n in sizes]
    want = [b"glTF" + hashlib.sha256(f).digest() * 3 for f in files]
    for rank, res, msz, mh, owned in got:
        own = dd.shard_indices(7, rank, 2, weights=sizes)
        assert owned == len(own) and len(res) == 7
        for i in range(7):
            if i in own:
                assert res[i][0] == want[i] and res[i][1] == [want[i][4:36], want[i][40:50]]
            else:
                assert res[i] is None                                                           # (not gathered)
        assert msz == [len(w) for w in want] and mh == [xxhash.xxh64(w).intdigest() for w in want]   # every rank holds the whole manifest
