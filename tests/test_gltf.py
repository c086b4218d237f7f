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
    doc2, bin2 = gltf.read_glb(out)
    assert "KHR_draco_mesh_compression" in doc2["extensionsRequired"]
    (payload, ids), = gltf.draco_blobs_of(out)
    assert payload[: len(want)] == want and len(payload) % 4 == 0 and len(payload) - len(want) < 4
    assert ids == {"NORMAL": 0, "POSITION": 1, "TEXCOORD_0": 2}
    p2 = doc2["meshes"][0]["primitives"][0]
    assert all("bufferView" not in doc2["accessors"][p2["attributes"][k]] for k in ids)
    assert len(out) < len(data)
