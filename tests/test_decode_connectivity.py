"""CPU tests of the connectivity decoder (draco-oxide_amd/csrc/dmi_decode_mesh.cpp, dmi_decode_connectivity): the header + connectivity
bytes the ENCODER wrote (bit-exact with the oracle: test_host_connectivity.py) must mean, by themselves, the tables that went in.  The reference
has no working connectivity decoder (decode/connectivity/spirale_reversi.rs:1088 is `unimplemented!`), so the statement is an isomorphism:
the encoder lists one corner per face in its coding order (`corners_of_edgebreaker`, edgebreaker.rs:549-560), the decoder lists its own in
the same order, and under that corner map every table (vertices, opposites, per-attribute vertices) must correspond one to one."""
import numpy as np
import pytest

import draco_oxide_amd as dmi
from draco_oxide_amd import synth
from helpers import obj_session, product_mesh_from_oracle
import test_gpu_parity as T


def _corner_map(enc_seeds, dec_seeds, nf):
    """encoder corner → decoder corner: seeds pair the faces up tip to tip; next/prev follow."""
    assert len(enc_seeds) == len(dec_seeds) == nf
    m = np.full(3 * nf, -1, np.int64)
    for k in range(3):
        e = 3 * (enc_seeds // 3) + (enc_seeds % 3 + k) % 3
        d = 3 * (dec_seeds // 3) + (dec_seeds % 3 + k) % 3
        m[e] = d
    assert (m >= 0).all() and len(np.unique(m)) == 3 * nf
    return m


def _same_partition(a, b):
    """two labelings of the same items induce the same partition"""
    pairs = np.unique(np.stack([a.astype(np.int64), b.astype(np.int64)], 1), axis=0)
    return len(pairs) == len(np.unique(a)) == len(np.unique(b))


def _by_point(att):
    return att.values if att.point_to_value is None else att.values[att.point_to_value]


def _round_trip(mesh, manifold=True):
    conn = dmi.encode_connectivity(mesh)
    try:
        dec = dmi.decode_connectivity(conn.bytes)
        assert dec["consumed"] == len(conn.bytes)
        nf = len(mesh.faces)
        assert len(dec["tables"]) == conn.num_tables
        m = _corner_map(conn.seeds().astype(np.int64), dec["seeds"].astype(np.int64), nf)
        inv = np.empty_like(m); inv[m] = np.arange(3 * nf)
        for i in range(conn.num_tables):
            e, d = conn.table(i), dec["tables"][i]
            assert d["num_faces"] == nf and d["num_vertices"] == e["num_vertices"], f"table {i}: vertex count"
            assert _same_partition(e["corner_to_vertex"], d["corner_to_vertex"][m]), f"table {i}: vertices"
            eo, do = e["opposite"].astype(np.int64), d["opposite"].astype(np.int64)
            none = 0xFFFFFFFF
            want = np.where(eo == none, none, m[np.minimum(eo, 3 * nf - 1)])
            assert (do[m] == want).all(), f"table {i}: opposites"
            # the left-most corner of every vertex is a corner of that vertex with nothing further left (open fans)
            lmc = d["left_most_corner"].astype(np.int64)
            assert (d["corner_to_vertex"][lmc] == np.arange(d["num_vertices"])).all()
        # points: corners the decoder calls one point were one point going in; a point the encoder's tables split into several vertices
        # (a non-manifold vertex: corner_table/mod.rs:418-470) comes back as several points, as from any Draco decoder
        ep, dp = conn.table(0)["corner_to_point"].astype(np.int64), dec["tables"][0]["corner_to_point"][m].astype(np.int64)
        pairs = np.unique(np.stack([ep, dp], 1), axis=0)
        assert len(pairs) == len(np.unique(dp)), "points"
        if manifold:
            assert len(pairs) == len(np.unique(ep)), "points of a manifold mesh"
    finally:
        conn.close()


@pytest.mark.parametrize("name", ["tetrahedron", "cube_quads", "sphere", "punctured_sphere", "torus"])
def test_fixture_connectivity_decodes_back(name):
    _round_trip(product_mesh_from_oracle(obj_session(name)))


@pytest.mark.parametrize("n,open_boundary,normals,uvs", [(3, False, True, True), (8, False, True, True), (17, True, True, True), (40, False, False, True), (33, True, True, False), (64, False, True, True)])
def test_synthetic_connectivity_decodes_back(n, open_boundary, normals, uvs):
    _round_trip(synth.torus_mesh(n, open_boundary=open_boundary, normals=normals, uvs=uvs))


@pytest.mark.parametrize("seed", range(12))
def test_random_soup_connectivity_decodes_back(seed):
    mesh, _ = T._soup_mesh(100 + seed, uv_per_corner=bool(seed & 1))
    _round_trip(mesh, manifold=False)


@pytest.mark.parametrize("n", [5, 23])
def test_heavy_tailed_connectivity_decodes_back(n):
    _round_trip(T._heavy_tailed_mesh(n, 7))


def test_handles_holes_and_components():
    # two components, one with holes punched in the middle (several boundary loops → topology splits), one closed with a handle
    a = synth.torus_mesh(14, open_boundary=True)
    faces = a.faces.reshape(-1, 3)
    keep = np.ones(len(faces), bool)
    keep[[40, 41, 42, 43, 120, 121, 200, 201, 202]] = False
    b = dmi.MeshBuilder()
    pos = _by_point(a.attributes[0])
    torus = synth.torus_mesh(9)
    tpos = _by_point(torus.attributes[0])
    all_pos = np.concatenate([pos, tpos + np.float32(10)])
    all_faces = np.concatenate([faces[keep], torus.faces.reshape(-1, 3) + len(pos)]).astype(np.uint32)
    b.add_attribute(all_pos, dmi.ATT_POSITION)
    b.set_connectivity_attribute(all_faces)
    _round_trip(b.build())


def _punched(n, frac, seed, open_boundary, uvs):
    faces, pos, nrm, uv = synth.torus_grid(n, open_boundary=open_boundary)
    rng = np.random.default_rng(seed)
    corner = faces[rng.random(len(faces)) > frac].ravel()
    b = dmi.MeshBuilder()
    pid = b.add_attribute(pos[corner], dmi.ATT_POSITION)
    if uvs:
        cuv = uv[corner].copy()
        cuv[np.repeat((np.arange(len(corner) // 3) % 5) == 0, 3)] += np.float32(0.25)   # seams
        b.add_attribute(cuv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, parents=[pid])
    b.set_connectivity_attribute(np.arange(len(corner), dtype=np.uint32).reshape(-1, 3))
    return b.build()


@pytest.mark.parametrize("seed", range(40))
def test_knocked_out_faces_topology_splits_and_seams(seed):
    """Random faces removed from a grid: boundary loops met mid-traversal (topology-split events, both orientations), handles, components that
    start at a boundary or inside; at 5-30 % removal a 1500-face mesh carries ~60-200 split events."""
    rng = np.random.default_rng(seed)
    mesh = _punched(int(rng.integers(4, 40)), float(rng.uniform(0, 0.4)), seed, bool(seed & 1), bool(seed & 2))
    if seed == 5:
        conn = dmi.encode_connectivity(_punched(30, 0.1, 2, True, False))
        from test_edgebreaker_kat import _parse_connectivity
        assert _parse_connectivity(conn.bytes)["splits"] > 20   # the construction does produce split events
        conn.close()
    _round_trip(mesh, manifold=False)


def test_malformed_connectivity_is_an_error_code():
    mesh = synth.torus_mesh(6)
    conn = dmi.encode_connectivity(mesh)
    good = conn.bytes
    conn.close()
    rng = np.random.default_rng(3)
    for cut in (0, 5, 11, 12, 20, len(good) // 2, len(good) - 1):
        with pytest.raises(dmi.DracoMiError):
            dmi.decode_connectivity(good[:cut])
    rejected = 0
    for _ in range(300):   # flipped bytes: an error code or some other mesh, never a crash
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(11, len(b)))] ^= 1 << int(rng.integers(0, 8))
        try:
            dmi.decode_connectivity(bytes(b))
        except dmi.DracoMiError:
            rejected += 1
    assert rejected > 0


def test_decode_budget_bounds_what_a_file_may_ask_for():
    """the whole-file decoder sizes per-table arrays from counts the FILE states (255 tables × millions of faces from a ~1 MB file);
    they are bounded by DMI_DECODE_BUDGET_MB before anything is allocated, and the call returns DMI_ERR_OUT_OF_MEMORY (11) — no abort."""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import draco_oxide_amd as dmi; from draco_oxide_amd import synth\n"
        "m = synth.torus_mesh(90)\n"                      # 16 200 faces, two attribute tables
        "conn = dmi.encode_connectivity(m); b = conn.bytes; conn.close()\n"
        "try:\n    dmi.decode_connectivity(b); print('decoded')\n"
        "except dmi.DracoMiError as e:\n    print('status', e.status)\n") % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, DMI_DECODE_BUDGET_MB="1", DMI_NO_TORCH_PREIMPORT="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300).stdout
    assert "status 11" in out, out
    env["DMI_DECODE_BUDGET_MB"] = "64"
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300).stdout
    assert "decoded" in out, out
