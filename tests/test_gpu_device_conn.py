"""The order-free half of the connectivity stage on the device (csrc/dmi_conn.hip) against the host builders (csrc/host_tables.cpp, both
pinned to the oracle by tests/test_host_connectivity.py): opposite corners, left-most corners, per-vertex boundary flags, vertex count —
identical arrays on closed / open grids, the OBJ fixtures, meshes with point → value maps; meshes outside the order-free class
(vertex-degenerate faces, edges with more than two faces, vertices with several fans) are flagged and take the reference's serial walks;
whole `.drc` bytes through dmi_encode_mesh with the device tables (default for ≥ 65 536 faces) and with DMI_HOST_CONNECTIVITY=1."""
import numpy as np
import pytest

import draco_oxide_amd as dmi
import orc
from draco_oxide_amd import synth
from helpers import obj_session, oracle_from_product_mesh, product_mesh_from_oracle
from test_gpu_parity import _assert_same, _soup_mesh

pytestmark = pytest.mark.gpu


def _host_tables(mesh):
    conn = dmi.encode_connectivity(mesh)
    t = conn.table(0)
    conn.close()
    return t


def _compare(mesh, what):
    want = _host_tables(mesh)
    got = dmi.device_corner_table(mesh)
    assert got["flags"] & (dmi.binding.CONN_DEGENERATE | dmi.binding.CONN_NONMANIFOLD_EDGE | dmi.binding.CONN_MULTI_FAN | dmi.binding.CONN_BAD_INDEX) == 0, (what, got["flags"])
    assert got["num_vertices"] == want["num_vertices"], what
    assert np.array_equal(got["opposite"], want["opposite"]), what
    assert np.array_equal(got["left_most_corner"], want["left_most_corner"]), what
    opp, lmc = want["opposite"], want["left_most_corner"]
    nxt = lambda c: np.where(c % 3 == 2, c - 2, c + 1)
    assert np.array_equal(got["on_boundary"].astype(bool), opp[nxt(lmc)] == 0xFFFFFFFF), what
    has_boundary = bool((opp == 0xFFFFFFFF).any())
    assert bool(got["flags"] & dmi.binding.CONN_HAS_BOUNDARY) == has_boundary, what


@pytest.mark.parametrize("n,open_boundary", [(3, False), (5, True), (40, False), (33, True), (150, False), (257, True), (400, False)])
def test_device_tables_equal_host_tables_on_grids(n, open_boundary):
    _compare(synth.torus_mesh(n, normals=False, uvs=False, open_boundary=open_boundary), f"grid {n} open={open_boundary}")


@pytest.mark.parametrize("n,labels,face_order", [(120, "split", "file"), (260, "split", "file"), (260, "random", "file"), (260, "file", "random"), (300, "stride", "random"),
                                                 (1100, "split", "file"), (1100, "file", "file")])
def test_device_tables_when_a_tile_of_faces_leaves_its_bucket_window(n, labels, face_order):
    """k_conn_faces ranks the half-edges of a tile of faces through an LDS window of 8192 buckets above the tile's smallest one; the others take the
    global atomic.  Vertex labels split into two far-apart halves / strided / random, and random face orders, put a tile's buckets on both sides of the
    window's end (and n = 1100: 2.4M faces, the 16-faces-per-thread tile)."""
    faces, pos, _, _ = synth.torus_grid(n, normals=False, uvs=False)
    V = n * n
    rng = np.random.default_rng(n)
    if labels == "split":        # even vertices first, odd ones V/2 further up
        perm = np.where(np.arange(V) % 2 == 0, np.arange(V) // 2, (V + 1) // 2 + np.arange(V) // 2)
    elif labels == "stride":     # v → v · 7919 mod V (7919 prime, no factor of n here)
        perm = (np.arange(V, dtype=np.int64) * 7919) % V
    elif labels == "random":
        perm = rng.permutation(V)
    else:
        perm = np.arange(V)
    assert len(np.unique(perm)) == V
    inv = np.empty(V, np.int64); inv[perm] = np.arange(V)
    f2 = perm[faces.astype(np.int64)].astype(np.uint32)
    if face_order == "random":
        f2 = f2[rng.permutation(len(f2))]
    mesh = dmi.Mesh(np.ascontiguousarray(f2), [dmi.Attribute(np.ascontiguousarray(pos[inv]), dmi.ATT_POSITION, dmi.DOMAIN_POSITION, unique_id=0)])
    _compare(mesh, f"grid {n} labels={labels} faces={face_order}")


@pytest.mark.parametrize("name", ["tetrahedron", "cube_quads", "sphere", "punctured_sphere", "torus"])
def test_device_tables_equal_host_tables_on_fixtures(name):
    _compare(product_mesh_from_oracle(obj_session(name)), name)   # position maps (value dedup) included


def test_device_tables_with_a_position_map():
    # every corner its own point, positions deduplicated by the builder: c2v comes from the point → value map
    faces, pos, nrm, uv = synth.torus_grid(30)
    cp = faces.ravel()
    b = dmi.MeshBuilder()
    b.add_attribute(pos[cp], dmi.ATT_POSITION)
    b.set_connectivity_attribute(np.arange(len(cp), dtype=np.uint32).reshape(-1, 3))
    mesh = b.build()
    assert mesh.attributes[0].point_to_value is not None
    _compare(mesh, "corner soup with a position map")


def test_meshes_outside_the_order_free_class_are_flagged_not_guessed():
    B = dmi.binding
    pos = np.random.default_rng(1).random((6, 3), dtype=np.float32)
    # three faces on the edge (0, 1)
    fan3 = dmi.Mesh(np.asarray([[0, 1, 2], [1, 0, 3], [0, 1, 4]], np.uint32), [dmi.Attribute(pos[:5], dmi.ATT_POSITION)])
    got = dmi.device_corner_table(fan3)
    assert got["num_vertices"] == 0 and got["flags"] & B.CONN_NONMANIFOLD_EDGE
    # two fans meeting in vertex 0 (a bow tie)
    bow = dmi.Mesh(np.asarray([[0, 1, 2], [0, 3, 4]], np.uint32), [dmi.Attribute(pos[:5], dmi.ATT_POSITION)])
    got = dmi.device_corner_table(bow)
    assert got["num_vertices"] == 0 and got["flags"] & B.CONN_MULTI_FAN
    # a vertex-degenerate face
    deg = dmi.Mesh(np.asarray([[0, 1, 2], [2, 2, 3]], np.uint32), [dmi.Attribute(pos[:4], dmi.ATT_POSITION)])
    got = dmi.device_corner_table(deg)
    assert got["num_vertices"] == 0 and got["flags"] & B.CONN_DEGENERATE
    # an unused vertex id below the largest one: the reference panics (corner_table/mod.rs:105-108) → error code
    unused = dmi.Mesh(np.asarray([[0, 1, 3]], np.uint32), [dmi.Attribute(pos[:4], dmi.ATT_POSITION)])
    with pytest.raises(dmi.DracoMiError):
        dmi.device_corner_table(unused)
    # a face index past the points
    bad = dmi.Mesh(np.asarray([[0, 1, 9]], np.uint32), [dmi.Attribute(pos[:4], dmi.ATT_POSITION)])
    with pytest.raises(dmi.DracoMiError):
        dmi.device_corner_table(bad)


@pytest.mark.parametrize("n,open_boundary,normals,uvs", [(200, False, True, True), (190, True, True, True), (256, False, False, False), (300, True, False, True)])
def test_encode_mesh_bytes_with_device_tables_and_with_host_tables(n, open_boundary, normals, uvs, monkeypatch):
    mesh = synth.torus_mesh(n, normals=normals, uvs=uvs, open_boundary=open_boundary)   # ≥ 65 536 faces: the device tables are the default
    want = oracle_from_product_mesh(mesh).encode()
    _assert_same(dmi.encode_mesh(mesh), want, f"grid {n}, device tables")
    monkeypatch.setenv("DMI_HOST_CONNECTIVITY", "1")
    _assert_same(dmi.encode_mesh(mesh), want, f"grid {n}, host tables")


def test_large_soup_falls_back_to_the_reference_walks():
    rng = np.random.default_rng(5)
    nv, nf = 30000, 70000
    pos = rng.random((nv, 3), dtype=np.float32)
    faces = rng.integers(0, nv, size=(nf, 3), dtype=np.uint32)
    faces = faces[(faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])]
    used = np.unique(faces)
    remap = np.full(nv, 0xFFFFFFFF, np.uint32); remap[used] = np.arange(len(used), dtype=np.uint32)
    mesh = dmi.Mesh(remap[faces], [dmi.Attribute(pos[used], dmi.ATT_POSITION)])
    try:
        want = oracle_from_product_mesh(mesh).encode()
    except orc.OracleError:
        pytest.skip("reference rejects this soup")
    _assert_same(dmi.encode_mesh(mesh), want, "random soup (non-manifold): host walks behind the device flags")


# ---- attribute corner tables on the device (k_att_*: attribute_corner_table.rs:16-137) ----
def _compare_attribute_table(mesh, att_index, what):
    """Device-built attribute table against the host builder's (itself pinned to the oracle and the reference's KATs)."""
    conn = dmi.encode_connectivity(mesh)
    uni, want = conn.table(0), conn.table(att_index)
    conn.close()
    got = dmi.device_attribute_table(mesh, att_index)
    assert got["num_vertices"] > 0, (what, got["flags"])
    own_table = not (np.array_equal(want["corner_to_vertex"], uni["corner_to_vertex"]) and np.array_equal(want["opposite"], uni["opposite"]))
    assert got["interior_seams"] == own_table, what
    if not own_table:   # no seam but the boundary: flags = the universal table's boundary corners, one attribute vertex per vertex
        assert np.array_equal(got["seam_edge"].astype(bool), uni["opposite"] == 0xFFFFFFFF), what
        assert got["num_vertices"] == uni["num_vertices"] and np.array_equal(got["corner_to_vertex"], uni["corner_to_vertex"]), what
        return got
    assert got["num_vertices"] == want["num_vertices"], what
    assert np.array_equal(got["corner_to_vertex"], want["corner_to_vertex"]), what
    assert np.array_equal(got["opposite"], want["opposite"]), what
    assert np.array_equal(got["left_most_corner"], want["left_most_corner"]), what
    assert np.array_equal(got["seam_edge"].astype(bool), want["opposite"] == 0xFFFFFFFF), what
    return got


def test_device_attribute_table_tetrahedron_kat():
    # core/corner_table/attribute_corner_table.rs:244-291 test_att_seam: the UV table of tetrahedron.obj
    mesh = product_mesh_from_oracle(obj_session("tetrahedron"))
    uv = [i for i, a in enumerate(mesh.attributes) if a.att_type == dmi.ATT_TEXCOORD][0]
    got = _compare_attribute_table(mesh, uv, "tetrahedron uv")
    assert all(got["seam_edge"][c] for c in [3, 5, 6, 7, 9, 11])
    assert got["left_most_corner"].tolist() == [6, 5, 11, 10, 8, 4] and got["corner_to_vertex"][0] == 0


@pytest.mark.parametrize("name", ["sphere", "torus", "punctured_sphere", "cube_quads"])
def test_device_attribute_tables_on_fixtures(name):
    mesh = product_mesh_from_oracle(obj_session(name))
    for i in range(1, len(mesh.attributes)):
        _compare_attribute_table(mesh, i, f"{name} attribute {i}")


@pytest.mark.parametrize("n", [7, 40, 129])
def test_device_attribute_tables_on_exporter_style_seams(n):
    faces, pos, nrm, uv = synth.seam_torus_rows(n, seed=77 + n)
    b = dmi.MeshBuilder()
    b.add_attribute(pos, dmi.ATT_POSITION)
    b.add_attribute(nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, parents=[0])
    b.add_attribute(uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, parents=[0])
    b.set_connectivity_attribute(faces)
    mesh = b.build()
    assert not _compare_attribute_table(mesh, 1, "normals repeat with the positions")["interior_seams"]
    assert _compare_attribute_table(mesh, 2, "uv seam along both closing curves")["interior_seams"]


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_device_attribute_tables_on_seam_soups(seed):
    """Random soups with per-corner UVs (seams nearly everywhere); meshes the universal kernels flag report no table (the host builds it)."""
    mesh, _ = _soup_mesh(seed, uv_per_corner=True)
    uni = dmi.device_corner_table(mesh)
    for i in range(1, len(mesh.attributes)):
        if uni["num_vertices"] == 0:
            got = dmi.device_attribute_table(mesh, i)
            assert got["num_vertices"] == 0 and got["flags"] != 0
        else:
            _compare_attribute_table(mesh, i, f"soup {seed} attribute {i}")


def test_seam_meshes_same_bytes_with_device_and_host_attribute_tables(monkeypatch):
    faces, pos, nrm, uv = synth.seam_torus_rows(60, seed=5)
    b = dmi.MeshBuilder()
    b.add_attribute(pos, dmi.ATT_POSITION)
    b.add_attribute(nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, parents=[0])
    b.add_attribute(uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, parents=[0])
    b.set_connectivity_attribute(faces)
    mesh = b.build()
    want = oracle_from_product_mesh(mesh).encode()

    def batch():
        jobs = dmi.meshes_prepare([mesh, mesh])
        try:
            return [j.header_and_connectivity + s for j, s in zip(jobs, dmi.jobs_encode(jobs))]
        finally:
            for j in jobs:
                j.close()
    assert batch() == [want, want]
    monkeypatch.setenv("DMI_NO_SEAM_MASKS", "1")          # the seam streams by the reference's walk from the back instead of the traversal's masks
    assert batch() == [want, want]
