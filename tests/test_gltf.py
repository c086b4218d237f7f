"""GLB container plumbing: primitive → Mesh exactly as the reference builds it, and the transcode round trip."""
import json
import os

import numpy as np
import pytest

import draco_oxide_amd as dmi
import orc
from draco_oxide_amd import gltf

DUCK = os.path.join(os.path.dirname(__file__), "golden", "data", "Duck.glb")


def _duck():
    data = open(DUCK, "rb").read()
    doc, binary = gltf.read_glb(data)
    prim = doc["meshes"][0]["primitives"][0]
    return data, doc, binary, prim


def _oracle_session(doc, binary, prim):
    names = sorted(k for k in prim["attributes"] if k in ("POSITION", "NORMAL", "TEXCOORD_0"))
    pos_id = names.index("POSITION")
    specs = []
    for n in names:
        rows = gltf._accessor_f32(doc, binary, prim["attributes"][n])
        ty = {"POSITION": orc.POSITION, "NORMAL": orc.NORMAL, "TEXCOORD_0": orc.TEXCOORD}[n]
        specs.append(dict(data=rows, type=ty, domain=orc.DOM_POSITION if n == "POSITION" else orc.DOM_CORNER, parents=[] if n == "POSITION" else [pos_id]))
    idx = gltf._accessor_indices(doc, binary, prim["indices"]).reshape(-1, 3)
    return orc.Session.from_arrays(idx, specs)


def test_duck_primitive_matches_reference_mesh_construction():
    # Duck.glb = 2399 vertices, 4212 faces, pos+nrm+uv; ids NORMAL=0, POSITION=1, TEXCOORD_0=2, Position swapped to slot 0
    data, doc, binary, prim = _duck()
    mesh, names = gltf.primitive_to_mesh(doc, binary, prim)
    assert names == ["NORMAL", "POSITION", "TEXCOORD_0"]
    assert len(mesh.faces) == 4212
    assert [a.att_type for a in mesh.attributes] == [dmi.ATT_POSITION, dmi.ATT_NORMAL, dmi.ATT_TEXCOORD]
    assert [a.unique_id for a in mesh.attributes] == [1, 0, 2]
    assert mesh.attributes[1].parent_index == 0 and mesh.attributes[2].parent_index == 0
    # the numpy MeshBuilder and the oracle's restated MeshBuilder agree on the built mesh
    sess = _oracle_session(doc, binary, prim)
    assert (mesh.faces == sess.faces()).all()
    for a, o in zip(mesh.attributes, sess.attributes()):
        assert a.unique_id == o["id"] and (a.values == o["data"]).all() and a.num_points == o["len"]
        assert (a.point_to_value is None) == (o["p2v"] is None)
        if o["p2v"] is not None:
            assert (a.point_to_value == o["p2v"]).all()


def test_glb_container_round_trip():
    data, doc, binary, prim = _duck()
    again = gltf.write_glb(doc, binary)
    doc2, bin2 = gltf.read_glb(again)
    assert doc2 == doc and bin2[: len(binary)] == binary and len(again) % 4 == 0


@pytest.mark.gpu
def test_duck_transcode_blob_bit_exact():
    data, doc, binary, prim = _duck()
    out, blobs = gltf.transcode_glb(data)
    want = _oracle_session(doc, binary, prim).encode()
    assert blobs[0] == want
    # (one device, one process: the blob is a view INTO the output file — the attribute section went from the library's buffer straight into it)
    assert isinstance(blobs[0], memoryview) and isinstance(out, memoryview) and bytes(blobs[0]) == want
    lo, hi = np.frombuffer(out, np.uint8).ctypes.data, np.frombuffer(out, np.uint8).ctypes.data + len(out)
    assert lo <= np.frombuffer(blobs[0], np.uint8).ctypes.data < hi                  # (a view INTO the output file, which is the library's memory)
    doc2, bin2 = gltf.read_glb(out)
    assert "KHR_draco_mesh_compression" in doc2["extensionsRequired"]
    (payload, ids), = gltf.draco_blobs_of(out)
    assert payload[: len(want)] == want and len(payload) % 4 == 0 and len(payload) - len(want) < 4
    assert ids == {"NORMAL": 0, "POSITION": 1, "TEXCOORD_0": 2}
    p2 = doc2["meshes"][0]["primitives"][0]
    assert all("bufferView" not in doc2["accessors"][p2["attributes"][k]] for k in ids)
    assert len(out) < len(data)


# ---- BASELINE configs[3]: lists of assets, one batch ---------------------------------------------------------------------------
def _pad4(b):
    return b + b"\0" * ((4 - len(b) % 4) % 4)


def _make_asset(prims, interleave=False):
    """A glTF document + one buffer holding `prims` = [dict(pos, nrm|None, uv|None, idx, feat|None, index_type)] as triangle
    primitives of one mesh (+ an untouched extra bufferView that must be carried over)."""
    buf = bytearray()
    views, accessors, gprims = [], [], []

    def add_view(data, stride=None, target=None):
        off = len(buf)
        buf.extend(_pad4(bytes(data)))
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        views.append(v)
        return len(views) - 1

    for p in prims:
        attrs = {}
        cols = [("POSITION", p["pos"], "VEC3")] + ([("NORMAL", p["nrm"], "VEC3")] if p.get("nrm") is not None else []) + \
               ([("TEXCOORD_0", p["uv"], "VEC2")] if p.get("uv") is not None else [])
        if interleave:
            rows = np.concatenate([c[1] for c in cols], axis=1).astype("<f4")
            stride = rows.shape[1] * 4
            vi = add_view(rows.tobytes(), stride=stride)
            off = 0
            for name, arr, ty in cols:
                accessors.append({"bufferView": vi, "byteOffset": off, "componentType": 5126, "count": len(arr), "type": ty})
                attrs[name] = len(accessors) - 1
                off += arr.shape[1] * 4
        else:
            for name, arr, ty in cols:
                vi = add_view(arr.astype("<f4").tobytes())
                accessors.append({"bufferView": vi, "componentType": 5126, "count": len(arr), "type": ty})
                attrs[name] = len(accessors) - 1
        if p.get("feat") is not None:
            ct, dt = p.get("feat_type", (5123, "<u2"))
            vi = add_view(p["feat"].astype(dt).tobytes())
            accessors.append({"bufferView": vi, "componentType": ct, "count": len(p["feat"]), "type": "SCALAR"})
            attrs["_FEATURE_ID_0"] = len(accessors) - 1
        ct, dt = {"u16": (5123, "<u2"), "u32": (5125, "<u4"), "u8": (5121, "u1")}[p.get("index_type", "u32")]
        vi = add_view(p["idx"].astype(dt).tobytes())
        accessors.append({"bufferView": vi, "componentType": ct, "count": p["idx"].size, "type": "SCALAR"})
        gprims.append({"attributes": attrs, "indices": len(accessors) - 1, "mode": 4})
    extra = add_view(b"carried over, byte for byte!")
    doc = {"asset": {"version": "2.0"}, "buffers": [{"byteLength": len(buf)}], "bufferViews": views, "accessors": accessors,
           "meshes": [{"primitives": gprims}], "nodes": [{"mesh": 0}], "scenes": [{"nodes": [0]}], "scene": 0,
           "images": [{"bufferView": extra, "mimeType": "image/png"}]}
    return doc, bytes(buf)


def _prim(n, seed, normals=True, uvs=True, feat=False, open_boundary=False, index_type="u32"):
    from draco_oxide_amd import synth
    faces, pos, nrm, uv = synth.torus_grid(n, seed=seed, normals=normals, uvs=uvs, open_boundary=open_boundary)
    d = dict(pos=pos, nrm=nrm, uv=uv, idx=faces.ravel(), index_type=index_type)
    if feat:
        d["feat"] = (np.arange(len(pos)) // 7 % 50).astype(np.uint32)
    return d


def _oracle_blob(p):
    """The `.drc` the reference would embed for primitive dict `p`: attributes in sorted-name order, ids in add order."""
    cols = [("NORMAL", p.get("nrm")), ("POSITION", p["pos"]), ("TEXCOORD_0", p.get("uv"))]
    cols = [(n, a) for n, a in cols if a is not None]
    pos_id = [n for n, _ in cols].index("POSITION")
    specs = []
    for n, a in cols:
        ty = {"POSITION": orc.POSITION, "NORMAL": orc.NORMAL, "TEXCOORD_0": orc.TEXCOORD}[n]
        specs.append(dict(data=a.astype(np.float32), type=ty, domain=orc.DOM_POSITION if n == "POSITION" else orc.DOM_CORNER, parents=[] if n == "POSITION" else [pos_id]))
    if p.get("feat") is not None:
        specs.append(dict(data=p["feat"].astype(np.uint32).reshape(-1, 1), type=orc.CUSTOM, domain=orc.DOM_CORNER))
    return orc.Session.from_arrays(p["idx"].reshape(-1, 3), specs).encode()


@pytest.mark.gpu
def test_transcode_a_list_of_assets_as_one_batch(tmp_path):
    """configs[3] through the driver: GLB bytes, a `.glb` path, a `.gltf` with an external `.bin`, a `.gltf` with a data URI — separate
    and interleaved vertex buffers, u16 / u32 indices, primitives without normals / UVs, a `_FEATURE_ID_0` set — all primitives of all
    files in ONE batch; every embedded blob equals the oracle's for that primitive, other bufferViews are carried over."""
    import base64
    files = [
        [_prim(9, 1), _prim(14, 2, normals=False, index_type="u16"), _prim(11, 3, uvs=False, open_boundary=True)],
        [_prim(20, 4, feat=True), _prim(8, 5, index_type="u16")],
        [_prim(16, 6, open_boundary=True)],
        [_prim(12, 7, feat=True, normals=False, uvs=False), _prim(10, 8)],
    ]
    sources = []
    doc, buf = _make_asset(files[0])
    sources.append(gltf.write_glb(doc, buf))                                       # GLB bytes
    doc, buf = _make_asset(files[1], interleave=True)
    (tmp_path / "b.glb").write_bytes(gltf.write_glb(doc, buf))
    sources.append(str(tmp_path / "b.glb"))                                        # .glb path
    doc, buf = _make_asset(files[2])
    doc["buffers"][0]["uri"] = "c.bin"
    (tmp_path / "c.bin").write_bytes(buf)
    (tmp_path / "c.gltf").write_text(json.dumps(doc))
    sources.append(str(tmp_path / "c.gltf"))                                       # .gltf + external .bin
    doc, buf = _make_asset(files[3], interleave=True)
    doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(buf).decode()
    (tmp_path / "d.gltf").write_text(json.dumps(doc))
    sources.append(str(tmp_path / "d.gltf"))                                       # .gltf + data URI
    sources.append(open(DUCK, "rb").read())
    results = gltf.transcode_files(sources)
    assert len(results) == 5
    for prims, (glb, blobs) in zip(files, results[:4]):
        assert len(blobs) == len(prims)
        embedded = gltf.draco_blobs_of(glb)
        for p, blob, (payload, ids) in zip(prims, blobs, embedded):
            want = _oracle_blob(p)
            assert blob == want
            assert payload[: len(want)] == want and len(payload) - len(want) < 4
            expect_names = sorted(n for n in ("NORMAL", "POSITION", "TEXCOORD_0") if p.get({"NORMAL": "nrm", "POSITION": "pos", "TEXCOORD_0": "uv"}[n]) is not None)
            expect_names += ["_FEATURE_ID_0"] if p.get("feat") is not None else []
            assert ids == {n: k for k, n in enumerate(expect_names)}
        doc2, bin2 = gltf.read_glb(glb)
        img = doc2["bufferViews"][doc2["images"][0]["bufferView"]]
        assert bin2[img["byteOffset"]: img["byteOffset"] + img["byteLength"]] == b"carried over, byte for byte!"
        assert "KHR_draco_mesh_compression" in doc2["extensionsRequired"] and len(glb) % 4 == 0
    data, doc, binary, prim = _duck()
    assert results[4][1][0] == _oracle_session(doc, binary, prim).encode()
    # the single-process multi-device path (every mesh on device 0 here) gives the same blobs
    again = gltf.transcode_files(sources, devices="all")
    assert [b for _, bl in again for b in bl] == [b for _, bl in results for b in bl]


@pytest.mark.gpu
def test_transcode_1024_primitives_in_one_batch():
    """64 assets × 16 primitives = 1024 meshes through transcode_files in one dmi_jobs_encode; a sample of blobs against the oracle."""
    rng = np.random.default_rng(5)
    files = [[_prim(int(rng.integers(6, 40)), seed=1000 + 16 * f + k, open_boundary=bool((f + k) % 3 == 0), index_type="u16" if k % 2 else "u32") for k in range(16)] for f in range(64)]
    sources = [gltf.write_glb(*_make_asset(prims, interleave=bool(i % 2))) for i, prims in enumerate(files)]
    results = gltf.transcode_files(sources)
    assert sum(len(blobs) for _, blobs in results) == 1024
    for f, k in [(0, 0), (5, 3), (17, 15), (33, 8), (63, 15), (40, 1), (21, 7), (9, 12)]:
        assert results[f][1][k] == _oracle_blob(files[f][k])


@pytest.mark.gpu
@pytest.mark.parametrize("seams", [False, True])
def test_full_size_files_through_the_transcoder_against_the_oracle(seams):
    """BASELINE configs[3] at its stated sizes: 72 GLB files of 2k–200k triangles each (synth.batch_glbs; seams=True: the exporter-style sheets with
    repeated positions / normals and a UV seam → device MeshBuilder merges, an attribute corner table of the UVs' own) through gltf.transcode_files —
    dmi_transcoder: device build, batched prepare, device chains, several stages — and a seeded sample of the embedded blobs, the LARGEST and
    the smallest file among them, byte for byte against the oracle's encode of the same accessors."""
    from draco_oxide_amd import synth
    glbs, total = synth.batch_glbs(72, seed=synth.SEED + (11 if seams else 5), seams=seams)
    assert total > 2_000_000
    tm = {}
    results = gltf.transcode_files(glbs, timings=tm)
    assert tm["primitives_built"] == 72 and len(results) == 72
    sizes = [gltf.primitive_weight(*(lambda d: (d, d["meshes"][0]["primitives"][0]))(gltf.read_glb(g)[0])) for g in glbs]
    assert max(sizes) > 120_000 and min(sizes) < 4_000
    rng = np.random.default_rng(2024)
    sample = {int(np.argmax(sizes)), int(np.argmin(sizes))} | {int(i) for i in rng.choice(72, size=6, replace=False)}
    for i in sorted(sample):
        doc, binary = gltf.read_glb(glbs[i])
        want = _oracle_session(doc, binary, doc["meshes"][0]["primitives"][0]).encode()
        assert bytes(results[i][1][0]) == want, (i, sizes[i])
        (payload, ids), = gltf.draco_blobs_of(results[i][0])
        assert payload[: len(want)] == want and len(payload) - len(want) < 4


@pytest.mark.gpu
@pytest.mark.parametrize("name,value", [("DMI_SPIN_WAITS", "1"), ("DMI_NO_STREAM_COPY", "1"), ("DMI_SMALL_HEAD", "1"), ("DMI_STAGE_RAMP", "8"), ("DMI_PREPARE_THREADS", "4"), ("DMI_STAGE_PRIMITIVES", "7")])
def test_transcoder_scheduling_switches_do_not_change_the_files(name, value, monkeypatch):
    """Round 6's scheduling / waiting / copying switches steer HOW a transcode runs — the runtime's own waits instead of the polling ones, plain memcpy
    into staging, the smallest files first, ramped stage sizes, four prepare threads, stages of seven primitives —, never what it writes."""
    from draco_oxide_amd import synth
    glbs, _ = synth.batch_glbs(48, seed=synth.SEED + 31, seams=(name in ("DMI_SPIN_WAITS", "DMI_PREPARE_THREADS")))
    monkeypatch.delenv(name, raising=False)
    want = gltf.transcode_files(glbs, copy=True)
    monkeypatch.setenv(name, value)
    assert gltf.transcode_files(glbs, copy=True) == want


@pytest.mark.gpu
def test_transcoder_walks_over_quad_ids_give_the_same_files(monkeypatch):
    """Round 6: a stage none of whose primitives has a point → value map has its opposite corners read back as 4·face + k ids (built_group_issue_tables) and its
    host walks run over those — the same files as with DMI_NO_QUAD=1, plain ones (the quad class) and exporter-style seams (never in it) alike."""
    from draco_oxide_amd import synth
    for seams in (False, True):
        glbs, _ = synth.batch_glbs(40, seed=synth.SEED + 23, seams=seams)
        monkeypatch.delenv("DMI_NO_QUAD", raising=False)
        a = gltf.transcode_files(glbs, copy=True)
        monkeypatch.setenv("DMI_NO_QUAD", "1")
        b = gltf.transcode_files(glbs, copy=True)
        assert a == b, seams
        monkeypatch.delenv("DMI_NO_QUAD")
        doc, binary = gltf.read_glb(glbs[7])
        assert a[7][1][0] == _oracle_session(doc, binary, doc["meshes"][0]["primitives"][0]).encode()


@pytest.mark.gpu
def test_one_process_several_devices_gives_the_one_device_files():
    """dmi_transcode_assets with a device LIST: one dmi_transcoder per entry, the least loaded one takes the next primitive, the files are written by
    the same library threads — no second interpreter, no gather.  devices=[0, 0] (two transcoders on the one GPU of the test box) must give the files
    of devices=[0], and a damaged asset must fail the call with the library's message, nothing left running."""
    from draco_oxide_amd import binding, synth
    glbs, total = synth.batch_glbs(40, lo=800, hi=60000, seed=31)
    rng = np.random.default_rng(4)
    files = [[_prim(int(rng.integers(6, 30)), seed=3000 + 8 * f + k, open_boundary=bool((f + k) % 4 == 0), index_type="u16" if k % 2 else "u32") for k in range(5)] for f in range(6)]
    sources = glbs + [gltf.write_glb(*_make_asset(prims, interleave=bool(i % 2))) for i, prims in enumerate(files)]
    one, st1 = binding.transcode_assets(sources, devices=[0])
    two, st2 = binding.transcode_assets(sources, devices=[0, 0])
    assert st1["devices"] == 1 and st2["devices"] == 2 and st1["primitives"] == st2["primitives"] == 40 + 30
    assert [bytes(g) for g, _ in one] == [bytes(g) for g, _ in two]
    assert [[bytes(b) for b in bl] for _, bl in one] == [[bytes(b) for b in bl] for _, bl in two]
    assert bytes(two[41][1][2]) == _oracle_blob(files[1][2])
    broken = bytearray(sources[3])
    broken[40:60] = b"x" * 20                                        # inside the JSON chunk
    with pytest.raises(dmi.DracoMiError, match="gltf"):
        binding.transcode_assets(sources[:3] + [bytes(broken)] + sources[4:], devices=[0, 0])
    truncated = sources[5][: len(sources[5]) - 4096]               # the BIN chunk cut short: an accessor reaches past its buffer
    with pytest.raises(dmi.DracoMiError):
        binding.transcode_assets([sources[0], truncated], devices=[0])
    again, _ = binding.transcode_assets(sources[:4], devices=[0])   # the library is in working order after the failures
    assert bytes(again[2][0]) == bytes(one[2][0])


def test_primitive_the_reference_cannot_encode_is_refused():
    doc, buf = _make_asset([_prim(6, 1)])
    prim = doc["meshes"][0]["primitives"][0]
    prim["attributes"]["COLOR_0"] = prim["attributes"]["NORMAL"]      # sorts before POSITION: the reference's parent ids go wrong
    with pytest.raises(ValueError):
        gltf.primitive_to_mesh(doc, buf, prim)


def test_feature_id_accessor_conversions():
    vals = np.array([0, 1, 255, 7, 300.9, -4.0, np.nan], np.float32)
    doc = {"accessors": [{"bufferView": 0, "componentType": 5126, "count": len(vals), "type": "SCALAR"}], "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 4 * len(vals)}]}
    got = gltf._accessor_u32_scalars(doc, vals.astype("<f4").tobytes(), 0)
    assert got.tolist() == [0, 1, 255, 7, 300, 0, 0]                    # Rust `as u32`: truncates, saturates at 0, NaN → 0


@pytest.mark.gpu
def test_pipelined_stages_give_the_same_files(monkeypatch):
    """Small stages (build + prepare of stage k+1 beside the encode of stage k) against one stage."""
    rng = np.random.default_rng(9)
    files = [[_prim(int(rng.integers(6, 30)), seed=2000 + 8 * f + k, open_boundary=bool((f + k) % 4 == 0), index_type="u16" if k % 2 else "u32") for k in range(8)] for f in range(6)]
    sources = [gltf.write_glb(*_make_asset(prims, interleave=bool(i % 2))) for i, prims in enumerate(files)]
    one = gltf.transcode_files(sources, pipeline=False)
    monkeypatch.setattr(gltf, "PIPELINE_TRIANGLES", 3000)
    tm = {}
    many = gltf.transcode_files(sources, pipeline=True, timings=tm)
    assert [g for g, _ in many] == [g for g, _ in one]
    assert tm["primitives_built"] == 48 and tm["build_s"] > 0 and tm["encode_s"] > 0
    # (that was the library's own stage loop, dmi_transcoder; the interpreter's stage threads give the same files)
    monkeypatch.setenv("DMI_TRANSCODE_PYTHON", "1")
    tm2 = {}
    python_loop = gltf.transcode_files(sources, pipeline=True, timings=tm2)
    assert [g for g, _ in python_loop] == [g for g, _ in one] and "build_kernels_ms" in tm2
    monkeypatch.delenv("DMI_TRANSCODE_PYTHON")


@pytest.mark.gpu
def test_transcoder_object_results_order_and_errors():
    """dmi_transcoder through its binding: primitives pushed in two slices, stages reported through the callback, results by push index — the
    blob of primitive i = header + connectivity ++ section = the whole-mesh encode of its built mesh; a primitive with a zero normal
    fails the run at finish() with the library's error, and nothing is left half-made."""
    prims = [_prim(int(n), seed=300 + i, index_type="u16" if i % 2 else "u32") for i, n in enumerate([9, 14, 6, 21, 11, 8, 17])]
    raws, meshes = [], []
    for p in prims:
        doc, buf = _make_asset([p])
        g = gltf.write_glb(doc, buf)
        d, b = gltf.read_glb(g)
        raws.append(gltf.primitive_to_raw(d, b, d["meshes"][0]["primitives"][0])[0])
        meshes.append(gltf.primitive_to_mesh(d, b, d["meshes"][0]["primitives"][0])[0])
    seen = []
    with dmi.Transcoder(None, sum(len(r.indices) // 3 for r in raws), len(raws), on_done=lambda first, count: seen.append((first, count)), stage_triangles=400) as t:
        t.push(raws[:3])
        t.push(raws[3:])
        t.finish()
        assert sorted(k for f, c in seen for k in range(f, f + c)) == list(range(len(raws))) and len(seen) > 1
        for i, m in enumerate(meshes):
            (head, section), nf, npts = t.result(i)
            assert bytes(head) + bytes(section) == dmi.encode_mesh(m) and nf == len(m.faces)
    bad = dmi.RawMesh()                                                   # a zero normal: the reference panics in the octahedral transform (geom.rs:45) — an error code here
    pid = bad.add_attribute(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], np.float32), dmi.ATT_POSITION)
    bad.add_attribute(np.array([[0, 0, 1], [0, 0, 0], [0, 1, 0], [1, 0, 0]], np.float32), dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, parents=[pid])
    bad.set_indices(np.array([0, 1, 2, 1, 3, 2], np.uint32))
    with dmi.Transcoder(None, 2, 2) as t:
        t.push([raws[0], bad])
        with pytest.raises(dmi.DracoMiError):
            t.finish()


@pytest.mark.gpu
def test_primitive_without_a_surviving_face_is_left_alone():
    """encode.rs:934-936: a primitive whose built mesh has no face (all its faces are degenerate once equal points are merged) is not
    compressed; its accessors keep their bufferViews."""
    flat = _prim(6, 3)
    flat["pos"] = np.zeros_like(flat["pos"])
    flat["nrm"] = np.tile(np.array([[0, 0, 1]], np.float32), (len(flat["pos"]), 1))
    flat["uv"] = np.zeros_like(flat["uv"])
    good = _prim(7, 4)
    (glb, blobs), = gltf.transcode_files([gltf.write_glb(*_make_asset([flat, good]))])
    assert blobs == [_oracle_blob(good)]
    doc, _ = gltf.read_glb(glb)
    p0, p1 = doc["meshes"][0]["primitives"]
    assert "extensions" not in p0 and "bufferView" in doc["accessors"][p0["attributes"]["POSITION"]]
    assert "KHR_draco_mesh_compression" in p1["extensions"] and "bufferView" not in doc["accessors"][p1["attributes"]["POSITION"]]


@pytest.mark.gpu
def test_transcoded_blobs_decode_back_to_the_input_triangles():
    """Round trip through the product's whole-file decoder (dmi_decode_mesh): every blob of a transcode — seam-free and exporter-style seams —
    decodes to the input's triangles (quantized position rows, labelling-free) with the UVs within half a quantization step."""
    from draco_oxide_amd import synth
    from test_gpu_decode import numpy_quantize
    from test_gpu_decode_mesh import _canonical_faces, _requantize
    cases = [synth.torus_grid(23, seed=31), synth.seam_torus_rows(19, seed=32), synth.torus_grid(17, seed=33, open_boundary=True)]
    glbs = []
    for faces, pos, nrm, uv in cases:
        glbs.append(gltf.write_glb(*_make_asset([dict(pos=pos, nrm=nrm, uv=uv, idx=faces.ravel(), index_type="u32")])))
    results = gltf.transcode_files(glbs)
    for (faces, pos, nrm, uv), (glb, blobs) in zip(cases, results):
        dm = dmi.decode_mesh(blobs[0])
        q, mn, rg = numpy_quantize(pos, 11)
        got_rows = _requantize(dm["attributes"][0]["values"], mn, rg, 11)[dm["faces"].astype(np.int64)]
        assert dm["faces"].shape == faces.shape
        assert (_canonical_faces(q[faces.astype(np.int64)]) == _canonical_faces(got_rows)).all()
        duv = [a for a in dm["attributes"] if a["att_type"] == dmi.ATT_TEXCOORD][0]["values"]
        step = float(max(uv.max(), 0) - min(uv.min(), 0)) / 1023
        # per corner: the decoded UV of the corner's point against the input UV of a corner with the same quantized position and a nearby UV
        assert duv.shape[1] == 2 and np.isfinite(duv).all() and duv.min() >= uv.min() - step and duv.max() <= uv.max() + step


# ---- the reference's remaining public fixtures (draco-oxide/tests/data: data files only) in the `.gltf` + external `.bin` form ----
DATA = os.path.join(os.path.dirname(__file__), "golden", "data")


def test_reference_gltf_fixtures_parse_like_their_glb():
    """Duck/Duck.gltf + Duck0.bin is the same asset as Duck.glb (its buffer = the head of the GLB's BIN chunk); Triangle.gltf + simpleTriangle.bin is one
    face with positions only and no `mode` (TRIANGLES by default)."""
    doc, bufs = gltf.load_document(os.path.join(DATA, "Duck", "Duck.gltf"))
    _, gdoc, gbin, _ = _duck()
    assert len(bufs) == 1 and bytes(bufs[0]) == bytes(gbin[: len(bufs[0])])
    assert doc["meshes"][0]["primitives"][0]["attributes"] == gdoc["meshes"][0]["primitives"][0]["attributes"]
    tdoc, tbufs = gltf.load_document(os.path.join(DATA, "Triangle.gltf"))
    prim = tdoc["meshes"][0]["primitives"][0]
    assert "mode" not in prim and list(prim["attributes"]) == ["POSITION"] and len(tbufs[0]) == 44
    assert gltf._accessor_indices(tdoc, tbufs[0], prim["indices"]).tolist() == [0, 1, 2]


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [[0], [0, 0]])
def test_reference_gltf_fixtures_through_the_native_transcoder(devices):
    """the reference's own `.gltf` + `.bin` assets (io/gltf/transcoder.rs:281-341 names private files; these are the public ones) through
    dmi_transcode_assets in its `json + buffers` form: Duck's blob = the blob of Duck.glb = the oracle's; Triangle (one face, positions only) = the oracle's."""
    from draco_oxide_amd import binding
    duck_gltf, tri_gltf = os.path.join(DATA, "Duck", "Duck.gltf"), os.path.join(DATA, "Triangle.gltf")
    sources = [duck_gltf, tri_gltf, open(DUCK, "rb").read()]
    results, st = binding.transcode_assets(gltf._native_assets(sources), devices=devices)
    assert len(results) == 3 and st["devices"] == len(devices) and st["primitives"] == 3
    data, doc, binary, prim = _duck()
    want_duck = _oracle_session(doc, binary, prim).encode()
    assert bytes(results[0][1][0]) == want_duck and bytes(results[2][1][0]) == want_duck
    tdoc, tbufs = gltf.load_document(tri_gltf)
    want_tri = _oracle_session(tdoc, tbufs[0], tdoc["meshes"][0]["primitives"][0]).encode()
    assert bytes(results[1][1][0]) == want_tri
    for k, want in ((0, want_duck), (1, want_tri)):
        (payload, ids), = gltf.draco_blobs_of(results[k][0])
        assert payload[: len(want)] == want and len(payload) - len(want) < 4
    assert gltf.draco_blobs_of(results[1][0])[0][1] == {"POSITION": 0}
    # the written Duck: still names its texture by uri (an external image is not the transcoder's to move), geometry accessors without bufferViews
    doc2, _ = gltf.read_glb(results[0][0])
    assert doc2["images"][0]["uri"] == "DuckCM.png" and "KHR_draco_mesh_compression" in doc2["extensionsRequired"]
    p2 = doc2["meshes"][0]["primitives"][0]
    assert all("bufferView" not in doc2["accessors"][a] for a in p2["attributes"].values())


# ---- accessors shared between primitives (multi-material meshes share POSITION; a LINES primitive beside a TRIANGLES one) ----
def _shared_accessor_asset():
    """One mesh, three primitives over ONE set of vertex accessors (POSITION / NORMAL / TEXCOORD_0): two triangle primitives with index accessors of their
    own (the two halves of a grid: a multi-material mesh) and a LINES primitive on the same POSITION accessor, which stays uncompressed."""
    p = _prim(12, 77)
    doc, buf = _make_asset([p])
    idx = p["idx"].reshape(-1, 3)
    half = len(idx) // 2
    buf = bytearray(buf)

    def add(data, ct):
        off = len(buf)
        buf.extend(_pad4(bytes(data)))
        doc["bufferViews"].append({"buffer": 0, "byteOffset": off, "byteLength": len(data)})
        doc["accessors"].append({"bufferView": len(doc["bufferViews"]) - 1, "componentType": ct, "count": len(data) // (4 if ct == 5125 else 2), "type": "SCALAR"})
        return len(doc["accessors"]) - 1
    i0, i1 = add(idx[:half].astype("<u4").tobytes(), 5125), add(idx[half:].astype("<u4").tobytes(), 5125)
    il = add(np.arange(8, dtype="<u2").tobytes(), 5123)
    att = doc["meshes"][0]["primitives"][0]["attributes"]
    doc["meshes"][0]["primitives"] = [{"attributes": dict(att), "indices": i0, "mode": 4, "material": 0}, {"attributes": dict(att), "indices": i1, "mode": 4, "material": 1},
                                      {"attributes": {"POSITION": att["POSITION"]}, "indices": il, "mode": 1}]
    doc["materials"] = [{}, {}]
    doc["buffers"][0]["byteLength"] = len(buf)
    halves = [dict(p, idx=idx[:half].ravel()), dict(p, idx=idx[half:].ravel())]
    return doc, bytes(buf), p, halves


def test_shared_accessors_get_private_copies_python_assembly():
    doc, buf, p, halves = _shared_accessor_asset()
    n_acc = len(doc["accessors"])
    prims = gltf._plan(doc)
    assert len(prims) == 2
    pos = prims[0][0]["attributes"]["POSITION"]
    gltf._privatize_accessors(doc, prims, [(b"x", 10, 20), (b"y", 11, 21)])
    a0, a1, line = doc["meshes"][0]["primitives"]
    assert line["attributes"]["POSITION"] == pos and "bufferView" in doc["accessors"][pos]           # the LINES primitive keeps the original, untouched
    assert a0["attributes"]["POSITION"] != pos and a1["attributes"]["POSITION"] != pos and a0["attributes"]["POSITION"] != a1["attributes"]["POSITION"]
    assert a0["attributes"]["NORMAL"] != a1["attributes"]["NORMAL"]                                   # shared by the two compressed ones only: one copy, one original
    assert len(doc["accessors"]) == n_acc + 2 + 1 + 1                                               # POSITION ×2 (three users), NORMAL, TEXCOORD_0 ×1 each
    assert a0["indices"] != a1["indices"] and len({a0["indices"], a1["indices"]}) == 2              # index accessors were private already


@pytest.mark.gpu
def test_primitives_that_share_accessors_native_and_python(monkeypatch):
    """Two compressed primitives and an uncompressed LINES primitive over one POSITION accessor: every compressed primitive ends up with placeholder
    accessors of its own (its own counts), the LINES primitive's POSITION keeps its bufferView and its bytes; blobs = the oracle's for each half; the native
    loop and the interpreter's give the same file."""
    doc, buf, p, halves = _shared_accessor_asset()
    src = gltf.write_glb(doc, buf)
    (glb, blobs), = gltf.transcode_files([src])
    assert [bytes(b) for b in blobs] == [_oracle_blob(h) for h in halves]
    doc2, bin2 = gltf.read_glb(glb)
    a0, a1, line = doc2["meshes"][0]["primitives"]
    acc = doc2["accessors"]
    for a, blob in ((a0, blobs[0]), (a1, blobs[1])):
        assert "KHR_draco_mesh_compression" in a["extensions"]
        assert all("bufferView" not in acc[i] for i in list(a["attributes"].values()) + [a["indices"]])
    assert set(a0["attributes"].values()).isdisjoint(a1["attributes"].values())
    lp = acc[line["attributes"]["POSITION"]]
    assert "extensions" not in line and "bufferView" in lp and lp["count"] == len(p["pos"])
    v = doc2["bufferViews"][lp["bufferView"]]
    assert bytes(bin2[v["byteOffset"]: v["byteOffset"] + v["byteLength"]]) == p["pos"].astype("<f4").tobytes()
    # each compressed primitive's counts are its own (the built meshes of the two halves have different point counts only if a half drops points: both keep
    # the points their faces use)
    for a, h in ((a0, halves[0]), (a1, halves[1])):
        assert acc[a["indices"]]["count"] == len(h["idx"]) and acc[a["attributes"]["POSITION"]]["count"] == len(np.unique(h["idx"]))
    monkeypatch.setenv("DMI_TRANSCODE_PYTHON", "1")
    (glb_py, blobs_py), = gltf.transcode_files([src])
    assert bytes(glb_py) == bytes(glb)


@pytest.mark.gpu
def test_native_transcoder_refuses_accessors_it_would_misread():
    """normalized-integer / quantized inputs (componentType ≠ FLOAT) for POSITION / NORMAL / TEXCOORD_0, a byteStride below the element size, an
    accessor that leaves its bufferView (but not its buffer), a duplicated attribute name: refused with the library's message."""
    from draco_oxide_amd import binding
    def broken(edit, raw_json=None):
        doc, buf = _make_asset([_prim(8, 3)])
        edit(doc)
        if raw_json is not None:
            js = raw_json(json.dumps(doc, separators=(",", ":")))
            return (js.encode(), [buf])
        return gltf.write_glb(doc, buf)
    cases = [
        lambda d: d["accessors"][0].update(componentType=5123),                                   # POSITION as UNSIGNED_SHORT (KHR_mesh_quantization)
        lambda d: d["accessors"][2].update(type="VEC3"),                                          # TEXCOORD_0 typed VEC3
        lambda d: d["bufferViews"][0].update(byteStride=8),                                        # stride below a VEC3 row
        lambda d: d["bufferViews"][0].update(byteLength=d["bufferViews"][0]["byteLength"] - 12),  # the accessor leaves its view
    ]
    for edit in cases:
        with pytest.raises(dmi.DracoMiError, match="gltf"):
            binding.transcode_assets([broken(edit)], devices=[0])
    dup = broken(lambda d: None, raw_json=lambda js: js.replace('"attributes":{"POSITION":0', '"attributes":{"POSITION":0,"POSITION":0', 1))
    with pytest.raises(dmi.DracoMiError, match="twice"):
        binding.transcode_assets([dup], devices=[0])
    ok, _ = binding.transcode_assets([broken(lambda d: None)], devices=[0])
    assert len(ok) == 1
