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
    declared = set(re.findall(r"^(?:int|void|void\*|const char\*|dmi_transcoder\*|uint32_t)\s+(dmi_[a-z_]+)\s*\(", hdr, flags=re.M))
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


@pytest.mark.parametrize("n,open_boundary", [(12, False), (17, True), (40, False), (120, False), (151, True), (190, False), (191, True)])   # (120 / 151: > 16384 seam flags, the all-zero stream coded by its period; 190 / 191: ≥ 2^16 faces, the traversal on stamps)
def test_connectivity_matches_oracle_on_synthetic(n, open_boundary):
    mesh = synth.torus_mesh(n, open_boundary=open_boundary)
    _compare_conn(oracle_from_product_mesh(mesh), mesh)


@pytest.mark.parametrize("n, open_boundary", [(725, False), (727, True)])
def test_connectivity_of_a_million_faces_matches_the_oracle(n, open_boundary):
    """From 2^16 faces the traversal keeps 32-bit stamps instead of byte flags and follows its previous loop with an L1 prefetch (host_conn.cpp Walker):
    same bytes, seeds and sequences as the oracle on a closed and an open 1.05M-face grid (positions only: the oracle's tables in a few seconds; the
    120- / 151-grids of the synthetic cases above and the fixtures cover both sides of the threshold)."""
    mesh = synth.torus_mesh(n, normals=False, uvs=False, open_boundary=open_boundary)
    assert len(mesh.faces) >= 1 << 20
    _compare_conn(oracle_from_product_mesh(mesh), mesh)


def _quad_case(mesh):
    """bytes, seeds and sequences of the walks over 4·face + k ids (DMI_TEST_QUAD: the form the device stage hands the walks of a mesh none of whose attributes
    needs a corner table of its own) against the same walks over 3·face + k ids; the table itself must hold the re-coded ids."""
    plain = dmi.encode_connectivity(mesh)
    os.environ["DMI_TEST_QUAD"] = "1"
    try:
        quad = dmi.encode_connectivity(mesh)
    finally:
        del os.environ["DMI_TEST_QUAD"]
    try:
        assert quad.bytes == plain.bytes
        assert (quad.seeds() == plain.seeds()).all()
        o3, o4 = plain.table(0)["opposite"], quad.table(0)["opposite"]
        some = o3 != 0xFFFFFFFF
        assert (o4[~some] == 0xFFFFFFFF).all() and (o4[some] == o3[some] + o3[some] // 3).all(), "the hook did not re-code the table: the case tests nothing"
        for i in range(len(mesh.attributes)):
            assert (quad.table(i)["sequence"] == plain.table(i)["sequence"]).all(), f"sequence {i}"
    finally:
        plain.close()
        quad.close()


@pytest.mark.parametrize("n,open_boundary", [(12, False), (17, True), (151, True), (190, False), (191, True)])   # (190 / 191: ≥ 2^16 faces, the walks on stamps)
def test_walks_over_quad_corner_ids_on_grids(n, open_boundary):
    _quad_case(synth.torus_mesh(n, open_boundary=open_boundary))
    _quad_case(synth.torus_mesh(n, normals=False, uvs=False, open_boundary=open_boundary))


@pytest.mark.parametrize("name", ["tetrahedron", "cube_quads", "sphere", "punctured_sphere", "torus"])
def test_walks_over_quad_corner_ids_on_fixtures(name):
    mesh = product_mesh_from_oracle(obj_session(name))
    pos = mesh.attributes[0]
    _quad_case(dmi.Mesh(mesh.faces, [pos]))   # (positions alone: the fixtures' other attributes have maps of their own — outside the form's class)


@pytest.mark.parametrize("sizes", [[(9, True), (14, False), (6, True), (30, False)], [(150, True), (120, False), (60, True), (100, False)]])
def test_walks_over_quad_corner_ids_with_many_components_and_splits(sizes):
    """Several components (one traversal each, interior and boundary start faces), holes (the boundary labelling and marking read the table through opp3),
    and a handle (S faces with topology splits); the larger set runs the walks on stamps."""
    rng = np.random.default_rng(11)
    parts, faces, base = [], [], 0
    for n, open_b in sizes:
        m = synth.torus_mesh(n, normals=False, uvs=False, open_boundary=open_b, seed=int(rng.integers(1 << 30)))
        parts.append(m.attributes[0].values)
        faces.append(m.faces + base)
        base += len(m.attributes[0].values)
    f = np.concatenate(faces)
    pos = np.concatenate(parts)
    # holes: faces taken out as long as every vertex keeps another face
    uses = np.bincount(f.ravel(), minlength=len(pos))
    drop = []
    for k in rng.permutation(len(f))[:400]:
        if (uses[f[k]] > 2).all() and len(drop) < 12:
            drop.append(int(k)); uses[f[k]] -= 1
    assert len(drop) == 12
    f = np.delete(f, drop, axis=0)
    mesh = dmi.Mesh(f.astype(np.uint32), [dmi.Attribute(pos, dmi.ATT_POSITION)])
    _compare_conn(oracle_from_product_mesh(mesh), mesh)
    _quad_case(mesh)


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


def _assert_built_like_oracle(mesh, sess):
    assert (mesh.faces == sess.faces()).all()
    atts = sess.attributes()
    assert len(mesh.attributes) == len(atts)
    for a, o in zip(mesh.attributes, atts):
        assert a.values.shape == o["data"].shape and a.values.tobytes() == o["data"].tobytes()
        assert (a.point_to_value is None) == (o["p2v"] is None)
        if o["p2v"] is not None:
            assert (a.point_to_value == o["p2v"]).all()
        assert a.num_points == o["len"] and a.unique_id == o["id"]


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_cpp_mesh_builder_matches_the_restated_builder(seed):
    """dmi_mesh_build (C++, hash based) against the oracle's MeshBuilder restatement on rows with repeated values, -0.0 / 0.0,
    NaNs, degenerate faces and points no face references (unreferenced-point removal, builder.rs:129-189)."""
    rng = np.random.default_rng(seed)
    n_pts, n_faces = 40, 30
    pool = rng.integers(-2, 3, size=(12, 3)).astype(np.float32)
    pos = pool[rng.integers(0, len(pool), size=n_pts)].copy()
    pos[3] = [-0.0, 1.0, 0.0]
    pos[9] = [0.0, 1.0, -0.0]
    if seed % 2 == 0:
        pos[5, 1] = np.nan
        pos[17, 1] = np.nan
    uv = (rng.integers(0, 3, size=(n_pts, 2)) / 2.0).astype(np.float32)
    nrm = pool[rng.integers(0, len(pool), size=n_pts)] + np.float32(0.25)
    ids = rng.integers(0, 4, size=(n_pts, 1)).astype(np.uint32)
    faces = rng.integers(0, n_pts - 5, size=(n_faces, 3)).astype(np.uint32)     # the last points stay unreferenced
    faces[4] = [7, 7, 2]                                                          # degenerate
    b = dmi.MeshBuilder()
    b.add_attribute(uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, parents=[1])       # Position is NOT added first: it must be swapped to slot 0
    pid = b.add_attribute(pos, dmi.ATT_POSITION)
    b.add_attribute(nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, parents=[pid])
    b.add_attribute(ids, dmi.ATT_CUSTOM, dmi.DOMAIN_CORNER)
    b.set_connectivity_attribute(faces)
    mesh = b.build()
    sess = orc.Session.from_arrays(faces, [dict(data=uv, type=orc.TEXCOORD, domain=orc.DOM_CORNER, parents=[1]), dict(data=pos, type=orc.POSITION),
                                           dict(data=nrm, type=orc.NORMAL, domain=orc.DOM_CORNER, parents=[1]), dict(data=ids, type=orc.CUSTOM, domain=orc.DOM_CORNER)])
    _assert_built_like_oracle(mesh, sess)
    assert mesh.attributes[0].att_type == dmi.ATT_POSITION and mesh.attributes[1].parent_index == 0


def test_byte_identical_nan_rows_merge_like_the_reference_hash():
    """core/mesh/builder.rs:254-279 keys a point on the BYTES of its unique values: byte-identical NaN rows merge although the value dedup keeps
    them apart.  Host builder and numpy builder against the oracle's literal restatement; expected point counts written out by hand."""
    from helpers import nan_twin_primitives, oracle_session_of_specs
    for name, specs, faces, n_points in nan_twin_primitives():
        b = dmi.MeshBuilder()
        for rows, t, d, par in specs:
            b.add_attribute(rows, t, d, parents=par)
        b.set_connectivity_attribute(faces)
        mesh = b.build()
        sess = oracle_session_of_specs(specs, faces)
        _assert_built_like_oracle(mesh, sess)
        assert mesh.attributes[0].num_points == n_points, name
        m2 = b.build_numpy()
        assert (m2.faces == mesh.faces).all(), name
        for a, c in zip(mesh.attributes, m2.attributes):
            assert a.values.tobytes() == c.values.tobytes() and a.num_points == c.num_points, name
            assert (a.point_to_value is None) == (c.point_to_value is None), name
            assert a.point_to_value is None or (a.point_to_value == c.point_to_value).all(), name


def test_cpp_and_numpy_builders_agree_when_every_point_is_referenced():
    faces, pos, nrm, uv = synth.torus_grid(12)
    corner = faces.ravel()
    b = dmi.MeshBuilder()
    pid = b.add_attribute(pos[corner], dmi.ATT_POSITION)
    b.add_attribute(nrm[corner], dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, parents=[pid])
    b.add_attribute(uv[corner], dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, parents=[pid])
    b.set_connectivity_attribute(np.arange(len(corner), dtype=np.uint32).reshape(-1, 3))
    m1, m2 = b.build(), b.build_numpy()
    assert (m1.faces == m2.faces).all()
    for a, c in zip(m1.attributes, m2.attributes):
        assert a.values.tobytes() == c.values.tobytes() and a.num_points == c.num_points
        assert (a.point_to_value is None) == (c.point_to_value is None)
        if a.point_to_value is not None:
            assert (a.point_to_value == c.point_to_value).all()


def _conn_snapshot(mesh):
    conn = dmi.encode_connectivity(mesh)
    snap = [conn.bytes, conn.seeds()]
    for i in range(conn.num_tables):
        t = conn.table(i)
        snap += [t["corner_to_vertex"], t["opposite"], t["left_most_corner"], t["sequence"], np.uint32(t["num_vertices"])]
    conn.close()
    return snap


def _same(a, b):
    return len(a) == len(b) and all((x == y) if isinstance(x, (bytes, np.integer)) else (x.shape == y.shape and (x == y).all()) for x, y in zip(a, b))


def _large_cases():
    """Meshes above the size at which the host stages go parallel (≥ 2^18 corners): closed / open grids, UV seams, a non-manifold
    soup (edge test → serial matching + edge breaking), two sheets glued at single vertices (manifold edges, several fans per vertex →
    serial left-most-corner walk with vertex splits), a position-degenerate face (→ serial matching)."""
    out = {}
    out["closed grid"] = synth.torus_mesh(420)             # (≥ 2^20 corners: the loops really run on several threads)
    out["open grid, no normals"] = synth.torus_mesh(310, normals=False, open_boundary=True)
    faces, pos, nrm, uv = synth.torus_grid(420, open_boundary=True)
    corner = faces.ravel()
    cuv = uv[corner].copy()
    cuv[np.repeat((np.arange(len(faces)) % 11) == 0, 3)] += np.float32(0.25)     # UV seams along scattered faces
    b = dmi.MeshBuilder()
    pid = b.add_attribute(pos[corner], dmi.ATT_POSITION)
    b.add_attribute(nrm[corner], dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, parents=[pid])
    b.add_attribute(cuv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, parents=[pid])
    b.set_connectivity_attribute(np.arange(len(corner), dtype=np.uint32).reshape(-1, 3))
    out["UV seams"] = b.build()
    rng = np.random.default_rng(7)
    n_pts = 40000
    sf = rng.integers(0, n_pts, size=(100000, 3)).astype(np.uint32)
    sf = sf[(sf[:, 0] != sf[:, 1]) & (sf[:, 1] != sf[:, 2]) & (sf[:, 2] != sf[:, 0])]
    used = np.unique(sf)
    remap = np.zeros(n_pts, np.uint32)
    remap[used] = np.arange(len(used), dtype=np.uint32)
    out["non-manifold soup"] = dmi.Mesh(remap[sf], [dmi.Attribute(rng.uniform(-1, 1, size=(len(used), 3)).astype(np.float32), dmi.ATT_POSITION)])
    f1, p1, _, _ = synth.torus_grid(220, open_boundary=True)
    f2 = f1.copy() + np.uint32(len(p1))
    p2 = p1 + np.float32(5.0)
    glue = f2.copy()
    for a_, b_ in ((0, 0), (777, 4242)):          # vertex b_ of sheet 2 becomes vertex a_ of sheet 1
        glue[glue == np.uint32(len(p1) + b_)] = np.uint32(a_)
    allp = np.concatenate([p1, p2])
    ff = np.concatenate([f1, glue])
    used = np.unique(ff)
    remap = np.zeros(len(allp), np.uint32)
    remap[used] = np.arange(len(used), dtype=np.uint32)
    out["sheets glued at vertices"] = dmi.Mesh(remap[ff], [dmi.Attribute(allp[used], dmi.ATT_POSITION)])
    f3, p3, _, _ = synth.torus_grid(300)
    p3 = p3.copy()
    p3[5] = p3[6]                                  # duplicate position value: the builder keeps both points (their faces become position-degenerate)
    b = dmi.MeshBuilder()
    pid = b.add_attribute(p3, dmi.ATT_POSITION)
    b.add_attribute(np.arange(len(p3), dtype=np.uint32).reshape(-1, 1), dmi.ATT_CUSTOM)   # keeps the two points distinct
    b.set_connectivity_attribute(f3)
    out["position-degenerate faces"] = b.build()
    return out


@pytest.mark.parametrize("name", ["closed grid", "open grid, no normals", "UV seams", "non-manifold soup", "sheets glued at vertices", "position-degenerate faces"])
def test_parallel_host_stages_equal_the_serial_walks_and_the_oracle(name, monkeypatch):
    """Large meshes build their corner tables on host threads when the result cannot depend on the corner order, and overlap the
    attribute tables / sequencers with the Edgebreaker walk; DMI_SERIAL_TABLES=1 forces the literal serial walks.  Same tables, same
    connectivity bytes, same seeds and sequences — and the oracle's."""
    mesh = _large_cases()[name]
    assert 3 * len(mesh.faces) >= 1 << 18
    monkeypatch.setenv("DMI_PARALLEL_TABLES", "1")   # (the library takes these builders from 2^21 corners by itself)
    fast = _conn_snapshot(mesh)
    monkeypatch.delenv("DMI_PARALLEL_TABLES")
    monkeypatch.setenv("DMI_SERIAL_TABLES", "1")
    slow = _conn_snapshot(mesh)
    assert _same(fast, slow)
    sess = oracle_from_product_mesh(mesh)
    try:
        sess.encode()
    except orc.OracleError:
        return   # (an input the reference cannot encode: the two forms agreeing is all there is to check)
    assert fast[0][11:] == bytes(sess.blob("conn.bytes"))


def test_recycled_host_arrays_do_not_change_the_bytes():
    """A large mesh takes its index arrays from the host pool (dmi_host.hpp): repeated calls, a release of everything the library
    keeps in between, and a different mesh in between give the same connectivity bytes, tables and sequences."""
    big = synth.torus_mesh(600)                       # 720k faces: above the pool's 4 MiB threshold
    other = synth.torus_mesh(450, open_boundary=True)

    def snapshot(mesh):
        conn = dmi.encode_connectivity(mesh)
        out = (bytes(conn.bytes), conn.seeds().copy(), [{k: (np.array(v).copy() if v is not None and not np.isscalar(v) else v) for k, v in conn.table(i).items()} for i in range(conn.num_tables)])
        conn.close()
        return out

    def same(a, b):
        assert a[0] == b[0] and (a[1] == b[1]).all() and len(a[2]) == len(b[2])
        for ta, tb in zip(a[2], b[2]):
            for k in ta:
                if isinstance(ta[k], np.ndarray):
                    assert (ta[k] == tb[k]).all(), k
                else:
                    assert ta[k] == tb[k], k

    first = snapshot(big)
    same(first, snapshot(big))                        # recycled arrays
    second = snapshot(other)
    same(first, snapshot(big))                        # arrays last used by a different mesh
    dmi.release_cached_memory()
    same(first, snapshot(big))
    same(second, snapshot(other))


def test_host_builder_against_the_restated_builder_on_random_primitives():
    """The generator of scripts/fuzz_build.py (duplicates, ±0.0, NaN rows, constant attributes, strided rows, degenerate faces, unreferenced points,
    Position not first) through dmi_mesh_build and the oracle's O(V²) restatement of MeshBuilder::build: 60 primitives of ≤ 250 points.  The device
    build is held to dmi_mesh_build on thousands of such primitives (tests/test_gpu_fuzz_slice.py, scripts/fuzz_build.py)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_build", os.path.join(root, "scripts", "fuzz_build.py"))
    fb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fb)
    otype = {dmi.ATT_POSITION: orc.POSITION, dmi.ATT_NORMAL: orc.NORMAL, dmi.ATT_TEXCOORD: orc.TEXCOORD, dmi.ATT_CUSTOM: orc.CUSTOM}
    done = 0
    for seed in range(4000, 4400):
        rng = np.random.default_rng(seed)
        specs, faces, _ = fb.random_primitive(rng)
        if len(specs[0][0]) > 250 or any(t not in otype for _, t, _, _ in specs):
            continue
        b = dmi.MeshBuilder()
        for rows, t, d, par in specs:
            b.add_attribute(rows, t, d, parents=par)
        b.set_connectivity_attribute(faces)
        sess = orc.Session.from_arrays(faces, [dict(data=np.ascontiguousarray(rows), type=otype[t], domain=orc.DOM_POSITION if d == dmi.DOMAIN_POSITION else orc.DOM_CORNER, parents=par)
                                               for rows, t, d, par in specs])
        try:
            mesh = b.build()
        except dmi.DracoMiError:
            continue
        _assert_built_like_oracle(mesh, sess)
        done += 1
        if done == 60:
            break
    assert done == 60


def test_host_calls_survive_a_fork_of_a_process_that_used_the_worker_pool():
    """Round 6: the library's short-lived workers are parked pool threads.  A forked child has none of them: it must start with an empty pool, not hand its
    tasks to threads that do not exist (host-only calls: the HIP runtime itself does not survive a fork)."""
    import subprocess, sys, os
    code = r"""
import os, sys
sys.path.insert(0, %r)
os.environ["DMI_NO_TORCH_PREIMPORT"] = "1"
import draco_oxide_amd as d
from draco_oxide_amd import synth
def host_work():
    faces, pos, nrm, uv = synth.torus_grid(600)
    b = d.MeshBuilder(); pid = b.add_attribute(pos, d.ATT_POSITION, d.DOMAIN_POSITION); b.add_attribute(uv, d.ATT_TEXCOORD, d.DOMAIN_CORNER, parents=[pid]); b.set_connectivity_attribute(faces)
    c = d.encode_connectivity(b.build()); n = bytes(c.bytes); c.close(); return n
a = host_work()
pid = os.fork()
if pid == 0:
    os._exit(0 if host_work() == a else 3)
_, st = os.waitpid(pid, 0)
sys.exit(os.WEXITSTATUS(st))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], timeout=300)
    assert r.returncode == 0
