"""Job creation relabels the connectivity inputs into coding order — with kernels for large meshes (csrc/dmi_relabel.hip), on host
threads for small ones.  DMI_RELABEL=device|host (read at job creation) forces a form; both must give the oracle's bytes on every
kind of table: seam-free, UV / normal seams (per-attribute tables with their own sequences), point_to_value maps, non-manifold soups,
open boundaries, high-valence fans, faces none of whose vertices is coded."""
import numpy as np
import pytest

import draco_oxide_amd as dmi
import orc
from draco_oxide_amd import synth
from helpers import obj_session, oracle_from_product_mesh, product_mesh_from_oracle, tables_from_oracle
from test_gpu_parity import _assert_same, _cones, _heavy_tailed_mesh, _soup_mesh

pytestmark = pytest.mark.gpu


def _both(mesh, want, what, monkeypatch, cfg=None):
    for mode in ("device", "host"):
        monkeypatch.setenv("DMI_RELABEL", mode)
        _assert_same(dmi.encode_mesh(mesh, cfg), want, f"{what} (relabelling: {mode})")
    monkeypatch.delenv("DMI_RELABEL")


@pytest.mark.parametrize("name", ["tetrahedron", "cube_quads", "sphere", "punctured_sphere", "torus"])
def test_fixtures_both_relabelling_forms(name, monkeypatch):
    sess = obj_session(name)
    want = sess.encode()
    _both(product_mesh_from_oracle(sess), want, name, monkeypatch)


@pytest.mark.parametrize("n,open_boundary,normals,uvs", [(5, False, True, True), (40, False, True, True), (33, True, True, True), (64, False, False, False), (150, True, False, True)])
def test_grids_both_relabelling_forms(n, open_boundary, normals, uvs, monkeypatch):
    mesh = synth.torus_mesh(n, normals=normals, uvs=uvs, open_boundary=open_boundary)
    _both(mesh, oracle_from_product_mesh(mesh).encode(), f"grid {n}", monkeypatch)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_soups_both_relabelling_forms(seed, monkeypatch):
    mesh, sess = _soup_mesh(seed, uv_per_corner=(seed % 2 == 0))
    try:
        want = sess.encode()
    except orc.OracleError:
        pytest.skip("reference rejects this soup")
    _both(mesh, want, f"soup {seed}", monkeypatch)
    monkeypatch.setenv("DMI_NO_FUSED", "1")
    _both(mesh, want, f"soup {seed}, per-attribute kernels", monkeypatch)


def test_seams_maps_custom_and_cones_both_relabelling_forms(monkeypatch):
    faces, pos, nrm, uv = synth.torus_grid(24)
    corner_pts = faces.ravel()
    cpos, cnrm, cuv = pos[corner_pts], nrm[corner_pts], uv[corner_pts].copy()
    band = (np.arange(len(faces)) % 7) == 0
    cuv[np.repeat(band, 3)] += np.float32(0.5)
    cnrm[::5] = cnrm[0]
    feat = (np.arange(len(corner_pts)) // 30).astype(np.uint32).reshape(-1, 1)
    b = dmi.MeshBuilder()
    pid = b.add_attribute(cpos, dmi.ATT_POSITION)
    b.add_attribute(cnrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, parents=[pid])
    b.add_attribute(cuv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, parents=[pid])
    b.add_attribute(feat, dmi.ATT_CUSTOM, dmi.DOMAIN_CORNER)
    f2 = np.arange(len(corner_pts), dtype=np.uint32).reshape(-1, 3)
    b.set_connectivity_attribute(f2)
    sess = orc.Session.from_arrays(f2, [dict(data=cpos, type=orc.POSITION), dict(data=cnrm, type=orc.NORMAL, domain=orc.DOM_CORNER, parents=[0]),
                                        dict(data=cuv, type=orc.TEXCOORD, domain=orc.DOM_CORNER, parents=[0]), dict(data=feat, type=orc.CUSTOM, domain=orc.DOM_CORNER)])
    _both(b.build(), sess.encode(), "seams + maps + custom", monkeypatch)
    for v, cs, w in ((9, True, False), (13, False, True), (400, False, False)):
        m = _cones(v, cs, w, seed=v)
        _both(m, oracle_from_product_mesh(m).encode(), f"cones {v}", monkeypatch)
    m = _heavy_tailed_mesh(48, seed=5)
    _both(m, oracle_from_product_mesh(m).encode(), "heavy tails", monkeypatch)


@pytest.mark.parametrize("name", ["sphere", "torus"])
def test_boundary_call_with_reference_tables_both_forms(name, monkeypatch):
    """dmi_encode_attributes fed with the oracle's tables / sequences (caller-owned arrays, no shared pointers between attribute
    tables: the alias detection compares contents), with and without caller-supplied sequences."""
    sess = obj_session(name)
    sess.encode()
    mesh = product_mesh_from_oracle(sess)
    want = bytes(sess.blob("atts.bytes"))
    for mode in ("device", "host"):
        monkeypatch.setenv("DMI_RELABEL", mode)
        tabs = tables_from_oracle(sess, len(mesh.attributes))
        _assert_same(dmi.encode_attributes(mesh.attributes, tabs), want, f"{name} boundary ({mode})")
        for t in tabs:
            t["sequence"] = None
        _assert_same(dmi.encode_attributes(mesh.attributes, tabs, seeds=sess.blob("conn.corners", np.uint32)), want, f"{name} boundary, library sequencer ({mode})")


def test_out_of_range_inputs_are_error_codes_in_both_forms(monkeypatch):
    """caller-supplied tables are range-checked (corner_to_vertex < num_vertices, opposite / left_most_corner / sequence /
    seeds inside [0, 3F), point_to_value < num_unique): error codes, never out-of-bounds accesses."""
    mesh = synth.torus_mesh(12)
    conn = dmi.encode_connectivity(mesh)
    base = [conn.table(i) for i in range(conn.num_tables)]
    seeds = conn.seeds()
    for mode in ("device", "host"):
        monkeypatch.setenv("DMI_RELABEL", mode)
        for field, bad in (("corner_to_vertex", 10 ** 6), ("opposite", 3 * len(mesh.faces)), ("left_most_corner", 3 * len(mesh.faces) + 7), ("sequence", 2 ** 31)):
            tabs = [dict(t) for t in base]
            arr = tabs[0][field].copy()
            arr[3] = bad
            tabs[0][field] = arr
            with pytest.raises(dmi.DracoMiError) as e:
                dmi.encode_attributes(mesh.attributes, tabs, seeds=seeds)
            assert e.value.status == 1, field
        bad_seeds = seeds.copy()
        bad_seeds[0] = 3 * len(mesh.faces)
        tabs = [dict(t, sequence=None) for t in base]
        with pytest.raises(dmi.DracoMiError):
            dmi.encode_attributes(mesh.attributes, tabs, seeds=bad_seeds)
        # a point_to_value entry past the unique values
        atts = list(mesh.attributes)
        p2v = np.arange(len(atts[1].values), dtype=np.uint32)
        p2v[5] = len(atts[1].values) + 3
        atts[1] = dmi.Attribute(atts[1].values, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, unique_id=1, parent_index=0, point_to_value=p2v)
        with pytest.raises(dmi.DracoMiError):
            dmi.encode_attributes(atts, [dict(t) for t in base], seeds=seeds)
    conn.close()


@pytest.mark.parametrize("n,tile,block", [(1, 64, 64), (63, 64, 64), (64, 64, 64), (1000, 64, 64), (5000, 256, 64), (70001, 16384, 16384), (100000, 4096, 128),
                                          (100000, 65536, 1024), (300007, 131072, 16384), (40000, 1 << 20, 256)])
def test_tile_sort_orders_every_tile_by_point(n, tile, block):
    """dmi_tile_sort_slots (k_tile_sort_local / k_tile_merge_global): inside every tile the slots hold the tile's points in ascending order and
    slot_entry is the permutation that goes with it — blocks that are whole tiles, tiles of several blocks (long strides in global memory), a
    partly filled last tile, a tile larger than the sequence."""
    rng = np.random.default_rng(n)
    s2p = rng.permutation(max(n, 1) * 3)[:n].astype(np.uint32)   # distinct points, as a sequence has them
    sp, se = dmi.tile_sort_slots(s2p, tile, block)
    t = 1 << int(np.ceil(np.log2(tile)))
    for a0 in range(0, n, t):
        a1 = min(n, a0 + t)
        assert (sp[a0:a1] == np.sort(s2p[a0:a1])).all(), f"tile at {a0}"
        assert ((se[a0:a1] >= a0) & (se[a0:a1] < a1)).all()
    assert (s2p[se] == sp).all() and len(np.unique(se)) == n
