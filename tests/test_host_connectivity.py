"""CPU tests of the product's host stages (corner tables, Edgebreaker, sequencer) against the oracle,
and of the C-ABI surface of libdraco_mi.so.  No GPU compute is invoked here."""
import ctypes
import os
import re

import numpy as np
import pytest

import draco_oxide_amd as dmi
import orc
from helpers import obj_session, oracle_from_product_mesh, product_mesh_from_oracle
from draco_oxide_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "draco_mi.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char\*)\s+(dmi_[a-z_]+)\s*\(", hdr, flags=re.M))
    assert declared >= set(dmi.binding.EXPORTS)
    L = ctypes.CDLL(dmi.library_path())
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/draco_mi.h but not exported"


def test_no_cpu_fallback_without_device():
    if dmi.device_count() > 0:
        pytest.skip("a GPU is present")
    mesh = synth.torus_mesh(8)
    with pytest.raises(dmi.DracoMiError) as e:
        dmi.encode_mesh(mesh)
    assert e.value.status == 9   # DMI_ERR_NO_DEVICE


def _compare_conn(sess, mesh):
    sess.encode()
    ref_conn = bytes(sess.blob("conn.bytes"))
    conn = dmi.encode_connectivity(mesh)
    assert conn.bytes[:11] == b"DRACO" + bytes([2, 2, 1, 1, 0, 0])
    assert conn.bytes[11:] == ref_conn
    assert (conn.seeds() == sess.blob("conn.corners", np.uint32)).all()
    t0 = conn.table(0)
    assert (t0["corner_to_vertex"] == sess.blob("ct.c2v", np.uint32)).all()
    assert (t0["opposite"] == sess.blob("ct.opp", np.uint32)).all()
    assert (t0["left_most_corner"] == sess.blob("ct.lmc", np.uint32)).all()
    for i in range(len(mesh.attributes)):
        t = conn.table(i)
        assert (t["sequence"] == sess.blob(f"att{i}.seq", np.uint32)).all(), f"sequence {i}"
        if i > 0:
            assert (t["corner_to_vertex"] == sess.blob(f"at{i-1}.c2v", np.uint32)).all()
            assert (t["opposite"] == sess.blob(f"at{i-1}.opp", np.uint32)).all()
            assert (t["left_most_corner"] == sess.blob(f"at{i-1}.lmc", np.uint32)).all()
    conn.close()


@pytest.mark.parametrize("name", ["tetrahedron", "cube_quads", "sphere", "punctured_sphere", "torus"])
def test_connectivity_matches_oracle_on_fixtures(name):
    sess = obj_session(name)
    _compare_conn(sess, product_mesh_from_oracle(sess))


@pytest.mark.parametrize("n,open_boundary", [(12, False), (17, True), (40, False)])
def test_connectivity_matches_oracle_on_synthetic(n, open_boundary):
    mesh = synth.torus_mesh(n, open_boundary=open_boundary)
    _compare_conn(oracle_from_product_mesh(mesh), mesh)


def test_connectivity_non_manifold_and_seams():
    rng = np.random.default_rng(5)
    # two fans sharing one vertex (non-manifold vertex), an edge with three faces, and UV seams
    pos = rng.uniform(-1, 1, size=(9, 3)).astype(np.float32)
    faces = np.array([[0, 1, 2], [0, 2, 3], [0, 4, 5], [0, 5, 6], [1, 2, 7], [2, 1, 8], [1, 2, 6]], np.uint32)
    uv_rows = rng.uniform(0, 1, size=(9, 2)).astype(np.float32)
    b = dmi.MeshBuilder()
    # expand to per-corner points so that UVs can differ per corner (seams)
    cpos = pos[faces.ravel()]
    cuv = uv_rows[faces.ravel()].copy()
    cuv[4] += np.float32(0.25)
    pid = b.add_attribute(cpos, dmi.ATT_POSITION)
    b.add_attribute(cuv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, parents=[pid])
    b.set_connectivity_attribute(np.arange(faces.size, dtype=np.uint32).reshape(-1, 3))
    mesh = b.build()
    sess = orc.Session.from_arrays(np.arange(faces.size, dtype=np.uint32).reshape(-1, 3),
                                   [dict(data=cpos, type=orc.POSITION), dict(data=cuv, type=orc.TEXCOORD, domain=orc.DOM_CORNER, parents=[0])])
    # the numpy MeshBuilder and the oracle's restated MeshBuilder must agree first
    assert (mesh.faces == sess.faces()).all()
    for a, o in zip(mesh.attributes, sess.attributes()):
        assert (a.values == o["data"]).all()
        assert (a.point_to_value is None) == (o["p2v"] is None)
        if o["p2v"] is not None:
            assert (a.point_to_value == o["p2v"]).all()
    _compare_conn(sess, mesh)


def test_mesh_builder_matches_oracle_on_fixture_rows():
    # rebuild sphere.obj from raw per-point rows through both builders
    s0 = obj_session("sphere")
    atts = s0.attributes()
    rows = [a["data"] if a["p2v"] is None else a["data"][a["p2v"]] for a in atts]
    b = dmi.MeshBuilder()
    pid = b.add_attribute(rows[0], dmi.ATT_POSITION)
    b.add_attribute(rows[1], dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, parents=[pid])
    b.set_connectivity_attribute(s0.faces())
    mesh = b.build()
    assert (mesh.faces == s0.faces()).all()
    for a, o in zip(mesh.attributes, atts):
        assert (a.values == o["data"]).all()
