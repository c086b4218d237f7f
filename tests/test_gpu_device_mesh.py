"""dmi_encode_mesh_device: encode::encode for a mesh whose faces, values and maps already live in HBM (what bench.py's `value` times).
Same bytes as dmi_encode_mesh from host memory and as the oracle — grids, a fixture with point → value maps, a mesh below the size from
which host-memory calls take the device tables, and a soup that falls back to the reference's serial walks (its values come down first)."""
import numpy as np
import pytest

import draco_oxide_amd as dmi
import orc
from draco_oxide_amd import synth
from helpers import obj_session, oracle_from_product_mesh, product_mesh_from_oracle
from test_gpu_parity import _assert_same, _soup_mesh

pytestmark = pytest.mark.gpu


def _check(mesh, what):
    want = oracle_from_product_mesh(mesh).encode()
    got = dmi.encode_mesh_device(dmi.DeviceMesh.upload(mesh))
    _assert_same(got, want, what + " (device-resident mesh)")
    assert got == dmi.encode_mesh(mesh), what


@pytest.mark.parametrize("n,open_boundary,normals,uvs", [(6, False, True, True), (60, True, True, True), (200, False, True, True), (260, True, False, True)])
def test_grids_from_hbm(n, open_boundary, normals, uvs):
    _check(synth.torus_mesh(n, normals=normals, uvs=uvs, open_boundary=open_boundary), f"grid {n}")


@pytest.mark.parametrize("name", ["sphere", "torus", "tetrahedron"])
def test_fixtures_with_maps_from_hbm(name):
    _check(product_mesh_from_oracle(obj_session(name)), name)


def test_soup_from_hbm_takes_the_host_walks():
    for seed in (1, 2, 3, 4):
        mesh, sess = _soup_mesh(seed, uv_per_corner=(seed % 2 == 0))
        try:
            want = sess.encode()
        except orc.OracleError:
            continue
        _assert_same(dmi.encode_mesh_device(dmi.DeviceMesh.upload(mesh)), want, f"soup {seed} (device-resident mesh)")


@pytest.mark.parametrize("n,normals,uvs,bits", [(200, True, True, None), (230, True, False, None), (210, False, True, None), (190, False, False, None), (220, True, True, (14, 12)),
                                                 (200, True, True, (22, 10)), (200, True, True, (11, 17))])
def test_early_quantization_before_the_walks_gives_the_same_bytes(n, normals, uvs, bits, monkeypatch):
    """A mesh in HBM of ≥ 2^16 faces has its value ranges and its quantization (value order, packed layouts) issued on a side stream BEFORE the host's
    serial walks; the pass then only gathers the packed values into coding order (k_seq_gather_packed).  Same bytes as the oracle's and as the call
    with the early stage off; the stage is entered exactly when the fused sweep's packed layouts apply (positions ≤ 21 bits, UVs ≤ 16 bits)."""
    mesh = synth.torus_mesh(n, normals=normals, uvs=uvs, seed=900 + n)
    assert len(mesh.faces) >= 1 << 16
    cfg = dmi.Config(flags=dmi.FLAG_TIMINGS) if bits is None else dmi.Config(pos_bits=bits[0], uv_bits=bits[1], flags=dmi.FLAG_TIMINGS)
    kw = {} if bits is None else dict(pos_bits=bits[0], uv_bits=bits[1])
    want = oracle_from_product_mesh(mesh).encode(**kw)
    dm = dmi.DeviceMesh.upload(mesh)
    got = dmi.encode_mesh_device(dm, cfg)
    entered = dmi.last_call_timings()["early_ms"] > 0
    _assert_same(got, want, f"early stage, grid {n} {kw}")
    packed_layouts = (normals or uvs) and (bits is None or (bits[0] <= 21 and (not uvs or bits[1] <= 16)))   # (positions alone are no fused sweep)
    assert entered == packed_layouts
    monkeypatch.setenv("DMI_NO_EARLY", "1")
    assert dmi.encode_mesh_device(dm, cfg) == got and dmi.last_call_timings()["early_ms"] == 0


@pytest.mark.parametrize("where", [0, 12345, -1])
def test_zero_normal_is_an_error_code_with_the_early_stage_too(where, monkeypatch):
    """geom.rs:45 asserts on a zero-length normal: an error code here.  Round 6: the early stage's quantizer looks for it (it reads the normals anyway; the
    range pass no longer does) and the first block of the record gather folds its per-block flags into the job's slot."""
    faces, pos, nrm, uv = synth.torus_grid(200, seed=77)
    nrm = nrm.copy()
    nrm[where] = 0
    atts = [dmi.Attribute(pos, dmi.ATT_POSITION), dmi.Attribute(nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, unique_id=1, parent_index=0),
            dmi.Attribute(uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, unique_id=2, parent_index=0)]
    dm = dmi.DeviceMesh.upload(dmi.Mesh(faces, atts))
    for early in (True, False):
        if not early:
            monkeypatch.setenv("DMI_NO_EARLY", "1")
        with pytest.raises(dmi.DracoMiError) as e:
            dmi.encode_mesh_device(dm, dmi.Config(flags=dmi.FLAG_TIMINGS))
        assert e.value.status == 6, early
    monkeypatch.delenv("DMI_NO_EARLY")
    nrm[where] = (0.0, 0.0, 1.0)   # the same mesh without the zero: through, and the early stage was entered
    atts[1] = dmi.Attribute(nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, unique_id=1, parent_index=0)
    mesh = dmi.Mesh(faces, atts)
    got = dmi.encode_mesh_device(dmi.DeviceMesh.upload(mesh), dmi.Config(flags=dmi.FLAG_TIMINGS))
    assert dmi.last_call_timings()["early_ms"] > 0
    _assert_same(got, oracle_from_product_mesh(mesh).encode(), "zero normal repaired")


def test_early_stage_is_dropped_when_an_attribute_has_seams_of_its_own():
    """The early stage guesses the fused sweep's layouts before any corner table exists; a UV attribute with interior seams leaves the sweep — the job's
    plan then differs from the guess, the early result is dropped and the job quantizes as always.  Same bytes as the oracle's."""
    faces, pos, nrm, uv = synth.seam_torus_rows(190, seed=5)
    b = dmi.MeshBuilder()
    pid = b.add_attribute(pos, dmi.ATT_POSITION)
    b.add_attribute(nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, parents=[pid])
    b.add_attribute(uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, parents=[pid])
    b.set_connectivity_attribute(faces)
    mesh = b.build()
    assert len(mesh.faces) >= 1 << 16
    want = oracle_from_product_mesh(mesh).encode()
    got = dmi.encode_mesh_device(dmi.DeviceMesh.upload(mesh), dmi.Config(flags=dmi.FLAG_TIMINGS))
    _assert_same(got, want, "seam torus from HBM")


@pytest.mark.parametrize("open_boundary,shuffle,normals,uvs", [(False, False, True, True), (True, False, True, True), (True, True, True, True), (False, True, True, False),
                                                               (True, False, False, True)])
def test_early_stage_on_irregular_meshes(open_boundary, shuffle, normals, uvs, monkeypatch):
    """The early stage (round 6: range partials folded by every block of the quantizer, the joint min/max by the first block of the record gather — no
    `_final` launches) on meshes of ≥ 2^16 faces with everything the sweep's fast path does not cover: hubs of valence 12 (fan-row overflow: the
    corner-table walk), valence 3 / 7 / 8, boundary fans and entries without a parallelogram (open grid), a scrambled point order (the value-order
    records are then gathered at random).  Same bytes as the oracle's and as the call without an early stage."""
    from helpers import irregular_grid_mesh
    mesh = irregular_grid_mesh(192, seed=17 + 2 * open_boundary + shuffle, open_boundary=open_boundary, shuffle_points=shuffle, normals=normals, uvs=uvs)
    assert len(mesh.faces) >= 1 << 16
    want = oracle_from_product_mesh(mesh).encode()
    dm = dmi.DeviceMesh.upload(mesh)
    cfg = dmi.Config(flags=dmi.FLAG_TIMINGS)
    got = dmi.encode_mesh_device(dm, cfg)
    _assert_same(got, want, f"early stage, irregular mesh, open={open_boundary} shuffled={shuffle}")
    assert dmi.last_call_timings()["early_ms"] > 0
    monkeypatch.setenv("DMI_NO_EARLY", "1")
    assert dmi.encode_mesh_device(dm, cfg) == got and dmi.last_call_timings()["early_ms"] == 0
