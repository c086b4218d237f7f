"""dmi_meshes_build — MeshBuilder::build on the device (core/mesh/builder.rs:62-90, core/attribute/mod.rs:394-452) —
against the host builder (dmi_mesh_build) and the oracle's restated builder: same unique-value order, maps, surviving points and faces;
then dmi_built_meshes_prepare on the resident result against dmi_meshes_prepare on the host meshes (same bytes)."""
import numpy as np
import pytest

import draco_oxide_amd as dmi
from draco_oxide_amd import synth
import orc
from helpers import obj_session

pytestmark = pytest.mark.gpu


def _same_mesh(a, b):
    assert a.faces.shape == b.faces.shape and (a.faces == b.faces).all()
    assert len(a.attributes) == len(b.attributes)
    for x, y in zip(a.attributes, b.attributes):
        assert x.values.shape == y.values.shape and x.values.tobytes() == y.values.tobytes()
        assert (x.point_to_value is None) == (y.point_to_value is None)
        if x.point_to_value is not None:
            assert (x.point_to_value == y.point_to_value).all()
        assert (x.num_points, x.unique_id, x.att_type, x.domain, x.parent_index) == (y.num_points, y.unique_id, y.att_type, y.domain, y.parent_index)


def _both(specs, faces, index_dtype=np.uint32):
    """(device-built Mesh, host-built Mesh) for attributes [(rows, type, domain, parents)] and faces."""
    rm, b = dmi.RawMesh(), dmi.MeshBuilder()
    for rows, t, d, par in specs:
        rm.add_attribute(rows, t, d, par)
        b.add_attribute(rows, t, d, parents=par)
    rm.set_indices(np.ascontiguousarray(faces, dtype=index_dtype).ravel())
    b.set_connectivity_attribute(faces)
    with dmi.meshes_build([rm], host_values=True) as batch:
        got = batch.mesh(0)
    return got, b.build()


def _messy(seed, n_pts=400, n_faces=500):
    rng = np.random.default_rng(seed)
    pool = rng.integers(-2, 3, size=(n_pts // 3, 3)).astype(np.float32)
    pos = pool[rng.integers(0, len(pool), size=n_pts)].copy()
    pos[3] = [-0.0, 1.0, 0.0]
    pos[9] = [0.0, 1.0, -0.0]
    if seed % 2 == 0:
        pos[5, 1] = np.nan
        pos[17, 1] = np.nan
        pos[18] = pos[17]
    uv = (rng.integers(0, 3, size=(n_pts, 2)) / 2.0).astype(np.float32)
    nrm = pool[rng.integers(0, len(pool), size=n_pts)] + np.float32(0.25)
    ids = rng.integers(0, 4, size=(n_pts, 1)).astype(np.uint32)
    faces = rng.integers(0, n_pts - 5, size=(n_faces, 3)).astype(np.uint32)   # the last points stay unreferenced
    faces[4] = [7, 7, 2]
    return pos, uv, nrm, ids, faces


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_device_build_matches_host_and_oracle_on_messy_rows(seed):
    pos, uv, nrm, ids, faces = _messy(seed)
    specs = [(uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, [1]), (pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION, []),
             (nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, [1]), (ids, dmi.ATT_CUSTOM, dmi.DOMAIN_CORNER, [])]
    got, want = _both(specs, faces)
    _same_mesh(got, want)
    assert dmi.last_build_timings()["device_meshes"] == 1
    sess = orc.Session.from_arrays(faces, [dict(data=uv, type=orc.TEXCOORD, domain=orc.DOM_CORNER, parents=[1]), dict(data=pos, type=orc.POSITION),
                                           dict(data=nrm, type=orc.NORMAL, domain=orc.DOM_CORNER, parents=[1]), dict(data=ids, type=orc.CUSTOM, domain=orc.DOM_CORNER)])
    assert (got.faces == sess.faces()).all()
    for a, o in zip(got.attributes, sess.attributes()):
        assert a.values.tobytes() == o["data"].tobytes()
        assert (a.point_to_value is None) == (o["p2v"] is None)
        if o["p2v"] is not None:
            assert (a.point_to_value == o["p2v"]).all()
        assert a.num_points == o["len"] and a.unique_id == o["id"]


def test_byte_identical_nan_rows_merge_on_the_device_too():
    """core/mesh/builder.rs:254-279 keys a point on the BYTES of its unique values (tests/helpers.py nan_twin_primitives): the device's byte
    classes against the host builder and the oracle's literal restatement, one primitive at a time and all of them in one batch."""
    from helpers import nan_twin_primitives, oracle_session_of_specs
    cases = nan_twin_primitives()
    raws = []
    for name, specs, faces, n_points in cases:
        got, want = _both(specs, faces)
        _same_mesh(got, want)
        assert dmi.last_build_timings()["device_meshes"] == 1, name
        assert got.attributes[0].num_points == n_points, name
        sess = oracle_session_of_specs(specs, faces)
        assert (got.faces == sess.faces()).all(), name
        for a, o in zip(got.attributes, sess.attributes()):
            assert a.values.tobytes() == o["data"].tobytes(), name
            assert (a.point_to_value is None) == (o["p2v"] is None), name
            assert o["p2v"] is None or (a.point_to_value == o["p2v"]).all(), name
            assert a.num_points == o["len"], name
        rm = dmi.RawMesh()
        for rows, t, d, par in specs:
            rm.add_attribute(rows, t, d, par)
        rm.set_indices(faces.ravel())
        raws.append(rm)
    with dmi.meshes_build(raws, host_values=True) as batch:
        assert dmi.last_build_timings()["device_meshes"] == len(cases)
        for j, (name, specs, faces, n_points) in enumerate(cases):
            assert batch.mesh(j).attributes[0].num_points == n_points, name


@pytest.mark.parametrize("index_dtype", [np.uint8, np.uint16, np.uint32])
def test_index_widths_and_strided_rows(index_dtype):
    n = 12 if index_dtype == np.uint8 else 40
    faces, pos, nrm, uv = synth.torus_grid(n)
    # an interleaved vertex buffer: pos | nrm | uv in 32-byte records (a glTF bufferView with byteStride 32)
    inter = np.zeros((len(pos), 8), np.float32)
    inter[:, 0:3], inter[:, 3:6], inter[:, 6:8] = pos, nrm, uv
    specs = [(inter[:, 0:3], dmi.ATT_POSITION, dmi.DOMAIN_POSITION, []), (inter[:, 3:6], dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, [0]), (inter[:, 6:8], dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, [0])]
    got, want = _both(specs, faces, index_dtype)
    _same_mesh(got, want)
    assert got.attributes[0].point_to_value is None and len(got.faces) == len(faces)


def test_corner_soup_merges_back_to_the_grid():
    """Every corner its own point (an unindexed primitive): value dedup + point merge must give back the indexed grid."""
    faces, pos, nrm, uv = synth.torus_grid(24)
    corner = faces.ravel()
    specs = [(pos[corner], dmi.ATT_POSITION, dmi.DOMAIN_POSITION, []), (nrm[corner], dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, [0]), (uv[corner], dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, [0])]
    got, want = _both(specs, np.arange(len(corner), dtype=np.uint32).reshape(-1, 3))
    _same_mesh(got, want)
    assert got.attributes[0].values.shape[0] == len(pos) and got.attributes[0].num_points == len(pos)


def test_all_points_identical_and_constant_attribute():
    """One value shared by every point of an attribute (every insert meets the same hash slot), positions distinct."""
    faces, pos, nrm, uv = synth.torus_grid(16)
    flat = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (len(pos), 1))
    zero = np.zeros((len(pos), 2), np.float32)
    zero[::2] = -0.0
    specs = [(pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION, []), (flat, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, [0]), (zero, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, [0])]
    got, want = _both(specs, faces)
    _same_mesh(got, want)
    assert got.attributes[1].values.shape[0] == 1 and got.attributes[2].values.shape[0] == 1


@pytest.mark.parametrize("name", ["tetrahedron", "sphere", "torus", "cube_quads", "punctured_sphere"])
def test_fixtures_rebuilt_from_rows(name):
    s0 = obj_session(name)
    atts = s0.attributes()
    rows = [a["data"] if a["p2v"] is None else a["data"][a["p2v"]] for a in atts]
    types = {orc.POSITION: dmi.ATT_POSITION, orc.NORMAL: dmi.ATT_NORMAL, orc.TEXCOORD: dmi.ATT_TEXCOORD}
    specs = [(rows[i], types[a["type"]], dmi.DOMAIN_POSITION if i == 0 else dmi.DOMAIN_CORNER, [] if i == 0 else [0]) for i, a in enumerate(atts)]
    got, want = _both(specs, s0.faces())
    _same_mesh(got, want)
    assert (got.faces == s0.faces()).all()


def test_batch_of_mixed_primitives_one_call():
    """Device-form and host-form primitives in one call, results in caller order; flags send meshes to the host builder."""
    rng = np.random.default_rng(7)
    raws, builders = [], []

    def add(specs, faces, dt=np.uint32):
        rm, b = dmi.RawMesh(), dmi.MeshBuilder()
        for rows, t, d, par in specs:
            rm.add_attribute(rows, t, d, par)
            b.add_attribute(rows, t, d, parents=par)
        rm.set_indices(np.ascontiguousarray(faces, dtype=dt).ravel())
        b.set_connectivity_attribute(faces)
        raws.append(rm)
        builders.append(b)

    for k in range(24):
        n = int(rng.integers(6, 30))
        faces, pos, nrm, uv = synth.torus_grid(n, seed=100 + k, open_boundary=bool(k % 3 == 0))
        specs = [(pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION, []), (nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, [0]), (uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, [0])]
        add(specs[: 1 + k % 3], faces, np.uint16 if k % 2 else np.uint32)
    pos, uv, nrm, ids, faces = _messy(11)
    add([(pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION, []), (ids, dmi.ATT_CUSTOM, dmi.DOMAIN_CORNER, [])], faces)
    # host-form: every face degenerate after the merge (no face survives), and attributes of different lengths
    same = np.zeros((6, 3), np.float32)
    add([(same, dmi.ATT_POSITION, dmi.DOMAIN_POSITION, [])], np.array([[0, 1, 2], [3, 4, 5]], np.uint32))
    add([(pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION, []), (uv[:-7], dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, [0])], faces)
    with dmi.meshes_build(raws, host_values=True) as batch:
        tm = dmi.last_build_timings()
        assert tm["device_meshes"] == 25 and tm["host_meshes"] == 2
        for j, b in enumerate(builders):
            _same_mesh(batch.mesh(j), b.build())


def test_twelve_attributes_stay_on_the_device_and_seventeen_go_to_the_host():
    """The device form takes up to 16 attributes per primitive (8 until round 4); more take the host builder inside the same call — same mesh."""
    rng = np.random.default_rng(12)
    faces, pos, nrm, uv = synth.torus_grid(20)
    corner = faces.ravel()[: 3 * 300]
    pos, nrm, uv = pos[corner], nrm[corner], uv[corner]
    tri = np.arange(len(corner), dtype=np.uint32).reshape(-1, 3)
    for extra, on_device in ((9, 1), (14, 0)):
        specs = [(pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION, []), (nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, [0]), (uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, [0])]
        specs += [(rng.integers(0, 3, size=(len(pos), 1)).astype(np.uint32), dmi.ATT_CUSTOM, dmi.DOMAIN_CORNER, []) for _ in range(extra)]
        got, want = _both(specs, tri)
        _same_mesh(got, want)
        tm = dmi.last_build_timings()
        assert tm["device_meshes"] == on_device and tm["host_meshes"] == 1 - on_device


def test_bad_index_goes_to_the_host_builder_and_errors_surface():
    faces, pos, nrm, uv = synth.torus_grid(8)
    bad = faces.copy()
    bad[3, 1] = len(pos) + 5
    got, want = _both([(pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION, [])], bad)
    _same_mesh(got, want)
    rm = dmi.RawMesh()
    rm.add_attribute(uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, [])
    rm.set_indices(faces.ravel())
    with pytest.raises(dmi.DracoMiError):
        dmi.meshes_build([rm])


def _raw_of(mesh):
    rm = dmi.RawMesh()
    for i, a in enumerate(mesh.attributes):
        rows = a.values if a.point_to_value is None else a.values[a.point_to_value]
        rm.add_attribute(rows, a.att_type, a.domain, [] if a.parent_index < 0 else [a.parent_index])
    rm.set_indices(mesh.faces.ravel())
    return rm


def test_built_prepare_equals_host_prepare_bytes():
    """dmi_meshes_build + dmi_built_meshes_prepare + dmi_jobs_encode against dmi_encode_mesh on the host-built meshes: same .drc bytes
    (closed / open grids, duplicated normals = a map on one attribute, a corner soup with UV seams → attribute table of its own)."""
    meshes = synth.batch_meshes(12, lo=500, hi=20000)
    raws = [_raw_of(m) for m in meshes]
    # duplicated normals on one mesh; a UV seam soup on another
    faces, pos, nrm, uv = synth.torus_grid(20)
    nrm2 = nrm.copy(); nrm2[1::2] = nrm2[0::2]
    b = dmi.MeshBuilder()
    rm = dmi.RawMesh()
    for rows, t, d, par in [(pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION, []), (nrm2, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, [0]), (uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, [0])]:
        b.add_attribute(rows, t, d, parents=par); rm.add_attribute(rows, t, d, par)
    b.set_connectivity_attribute(faces); rm.set_indices(faces.ravel())
    meshes.append(b.build()); raws.append(rm)
    corner = faces.ravel()
    cuv = uv[corner].copy()
    cuv[::7] += np.float32(0.125)                      # seams: the same position carries different UVs at different corners
    b, rm = dmi.MeshBuilder(), dmi.RawMesh()
    for rows, t, d, par in [(pos[corner], dmi.ATT_POSITION, dmi.DOMAIN_POSITION, []), (cuv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, [0])]:
        b.add_attribute(rows, t, d, parents=par); rm.add_attribute(rows, t, d, par)
    soup = np.arange(len(corner), dtype=np.uint32).reshape(-1, 3)
    b.set_connectivity_attribute(soup); rm.set_indices(soup.ravel())
    meshes.append(b.build()); raws.append(rm)
    want = [dmi.encode_mesh(m) for m in meshes]
    for host_values in (False, True):
        with dmi.meshes_build(raws, host_values=host_values) as batch:
            jobs = dmi.built_meshes_prepare(batch)
            try:
                sections = dmi.jobs_encode(jobs)
                got = [j.header_and_connectivity + s for j, s in zip(jobs, sections)]
            finally:
                for j in jobs:
                    j.close()
        assert got == want
    # a subset of a built batch, out of order
    with dmi.meshes_build(raws) as batch:
        pick = [5, 0, 13, 9]
        jobs = dmi.built_meshes_prepare(batch, which=pick)
        try:
            sections = dmi.jobs_encode(jobs)
            assert [j.header_and_connectivity + s for j, s in zip(jobs, sections)] == [want[i] for i in pick]
        finally:
            for j in jobs:
                j.close()


def test_large_mesh_build_and_prepare():
    """One mesh above the single-mesh threshold (2^20 faces): its own build group, prepared through the single-mesh path."""
    mesh = synth.torus_mesh(740)                        # 1 095 200 triangles
    with dmi.meshes_build([_raw_of(mesh)]) as batch:
        assert batch.summary(0)[0] == len(mesh.faces)
        job, = dmi.built_meshes_prepare(batch)
        try:
            got = job.header_and_connectivity + job.encode()
        finally:
            job.close()
    assert got == dmi.encode_mesh(mesh)


def test_exporter_style_seams_through_the_deferred_batch_path():
    """Meshes the way exporters write them (synth.seam_torus_rows: positions / normals repeated along the closing curves, texture coordinates
    with a seam there): MeshBuilder merges the positions, the UV attribute keeps a corner table of its own — built, prepared (deferred job
    creation with the seam table uploaded beside the sequences) and coded as a batch; bytes against the oracle's whole pipeline."""
    raws, want = [], []
    for k, n in enumerate((9, 14, 23, 31)):
        faces, pos, nrm, uv = synth.seam_torus_rows(n, seed=500 + k)
        rm = dmi.RawMesh()
        rm.add_attribute(pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION)
        rm.add_attribute(nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, [0])
        rm.add_attribute(uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, [0])
        rm.set_indices(faces.ravel())
        raws.append(rm)
        sess = orc.Session.from_arrays(faces, [dict(data=pos, type=orc.POSITION), dict(data=nrm, type=orc.NORMAL, domain=orc.DOM_CORNER, parents=[0]),
                                               dict(data=uv, type=orc.TEXCOORD, domain=orc.DOM_CORNER, parents=[0])])
        want.append(sess.encode())
    with dmi.meshes_build(raws, host_values=True) as batch:
        for j in range(len(raws)):
            m = batch.mesh(j)
            assert m.attributes[0].point_to_value is not None and m.attributes[2].point_to_value is None   # positions merged, UVs all distinct
        jobs = dmi.built_meshes_prepare(batch)
        try:
            got = [j.header_and_connectivity + s for j, s in zip(jobs, dmi.jobs_encode(jobs))]
        finally:
            for j in jobs:
                j.close()
        host_meshes = [batch.mesh(j) for j in range(len(raws))]
    assert got == want
    # the same meshes from host memory through dmi_meshes_prepare (host-packed groups defer their seam tables too)
    jobs = dmi.meshes_prepare(host_meshes)
    try:
        assert [j.header_and_connectivity + s for j, s in zip(jobs, dmi.jobs_encode(jobs))] == want
    finally:
        for j in jobs:
            j.close()


def test_rows_crafted_to_collide_go_to_the_host_builder():
    """Rows chosen so that thousands of them start at ONE slot of the value hash table: the kernels' probe cap flags the mesh (MB_CROWDED) and the
    host builder takes it — same mesh, no runaway kernel."""
    def mix(h, w):
        h = (h ^ w) & 0xFFFFFFFF
        h = (h * 0x85EBCA6B) & 0xFFFFFFFF
        h ^= h >> 13
        h = (h * 0xC2B2AE35) & 0xFFFFFFFF
        return h ^ (h >> 16)
    n_pts = 6000
    mask = 16384 - 1                                     # the table of 6000 points: the next power of two ≥ 12000
    cand = np.arange(1, 1 << 27, dtype=np.uint64)
    h = (np.uint64(0x9E3779B9) ^ cand) & np.uint64(0xFFFFFFFF)
    h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    hit = cand[(h & np.uint64(mask)) == 77][:n_pts].astype(np.uint32)
    assert len(hit) == n_pts and mix(0x9E3779B9, int(hit[5])) & mask == 77
    ids = hit.reshape(-1, 1)                              # 6000 distinct one-word rows, all starting at slot 77
    faces, pos, _, _ = synth.torus_grid(78)               # 6084 points
    pos, faces = pos[:n_pts], faces[(faces < n_pts).all(axis=1)]
    specs = [(pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION, []), (ids, dmi.ATT_CUSTOM, dmi.DOMAIN_CORNER, [])]
    got, want = _both(specs, faces)
    _same_mesh(got, want)
    tm = dmi.last_build_timings()
    assert tm["device_meshes"] == 0 and tm["host_meshes"] == 1
