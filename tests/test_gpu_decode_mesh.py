"""dmi_decode_mesh: a whole `.drc` the library wrote, read back from its bytes alone (header → Edgebreaker connectivity → attribute
section; draco-oxide_amd/csrc/dmi_decode_mesh.cpp).  Three statements per mesh:
  1. the decoded mesh IS the input mesh: the multiset of triangles, each as its three corners' quantized (position, uv) rows up to
     rotation, equals the input's under an independent numpy quantizer — no corner map, no table from the encoder involved;
  2. under the corner map the two coding orders give (tests/test_decode_connectivity.py), every attribute value equals — bit for bit —
     what dmi_decode_attributes returns when it is handed the ENCODER's tables (that path is pinned against the oracle's decoder in
     tests/test_gpu_decode.py);
  3. malformed files are error codes.
Same-author evidence (the reference has no working decoder: decode/ is not compiled, lib.rs:14), stated as such in DESIGN §2."""
import numpy as np
import pytest

import draco_oxide_amd as dmi
from draco_oxide_amd import synth
from test_decode_connectivity import _corner_map
from test_gpu_decode import numpy_quantize, _expand

pytestmark = pytest.mark.gpu


def _canonical_faces(rows_per_corner):
    """[F, 3, k] integer rows → every face as the lexicographically smallest of its three rotations, faces sorted: a labelling-free form."""
    f = np.asarray(rows_per_corner, np.int64)
    F, _, k = f.shape
    rots = np.stack([np.concatenate([f[:, (r + j) % 3] for j in range(3)], axis=1) for r in range(3)], axis=1)   # [F, 3, 3k]
    best = rots[:, 0].copy()
    for r in (1, 2):
        cand = rots[:, r]
        diff = cand != best
        first = np.argmax(diff, axis=1)
        less = diff.any(axis=1) & (cand[np.arange(F), first] < best[np.arange(F), first])
        best[less] = cand[less]
    return best[np.lexsort(best.T[::-1])]


def _requantize(values, mn, rng, bits):
    return np.rint((values.astype(np.float64) - mn.astype(np.float64)) / float(rng) * ((1 << bits) - 1)).astype(np.int64) if rng else np.zeros(values.shape, np.int64)


def _check_mesh(mesh, cfg=None, pos_bits=11, uv_bits=10):
    cfg = cfg or dmi.Config()
    drc = dmi.encode_mesh(mesh, cfg)
    dec = dmi.decode_mesh(drc)
    nf = len(mesh.faces)
    assert dec["faces"].shape == (nf, 3) and len(dec["attributes"]) == len(mesh.attributes)
    assert dec["faces"].max() == dec["num_points"] - 1
    in_faces = np.asarray(mesh.faces, np.int64).reshape(-1, 3)
    # 1. the mesh itself, labelling-free
    want_rows, got_rows = [], []
    for att, d in zip(mesh.attributes, dec["attributes"]):
        assert d["att_type"] == att.att_type and d["num_components"] == att.values.shape[1] and d["unique_id"] == att.unique_id
        assert d["values"].shape == (dec["num_points"], att.values.shape[1])
        if d["portabilization"] != 2:
            continue
        bits = d["bits"]
        assert bits == (pos_bits if att.att_type == dmi.ATT_POSITION else uv_bits if att.att_type == dmi.ATT_TEXCOORD else bits)
        q, mn, rng = numpy_quantize(att.values, bits)
        q = q if att.point_to_value is None else q[att.point_to_value]
        want_rows.append(q[in_faces])
        got_rows.append(_requantize(d["values"], mn, rng, bits)[dec["faces"].astype(np.int64)])
    assert (_canonical_faces(np.concatenate(want_rows, axis=2)) == _canonical_faces(np.concatenate(got_rows, axis=2))).all(), "the decoded triangles are not the input's"
    # 2. value for value against the table-given decoder
    conn = dmi.encode_connectivity(mesh)
    tables = [conn.table(i) for i in range(conn.num_tables)]
    seeds = conn.seeds()
    n_hdr = len(conn.bytes)
    conn.close()
    ref = dmi.decode_attributes(drc[n_hdr:], tables, mesh.attributes[0].num_points, seeds=seeds)
    dconn = dmi.decode_connectivity(drc)
    assert dconn["consumed"] == n_hdr
    m = _corner_map(seeds.astype(np.int64), dconn["seeds"].astype(np.int64), nf)
    dec_point_of_input_corner = dec["faces"].ravel().astype(np.int64)[m]
    for i, (r, d) in enumerate(zip(ref, dec["attributes"])):
        a = r["values"][in_faces.ravel()]
        b = d["values"][dec_point_of_input_corner]
        assert (a.view(np.uint32) == b.view(np.uint32)).all(), f"attribute {i}: decode_mesh and decode_attributes differ"
    # normals: within the 8-bit octahedral grid's error of the input, for nearly all of them (the reference's diamond inversion)
    for att, d in zip(mesh.attributes, dec["attributes"]):
        if d["portabilization"] == 3:
            n = _expand(att)[in_faces.ravel()]
            n = n / np.linalg.norm(n, axis=1, keepdims=True)
            cos = (n * d["values"][dec_point_of_input_corner]).sum(axis=1)
            generic = (np.abs(n) > 1e-3).all(axis=1)
            assert ((cos < np.cos(np.radians(2.5))) & generic).sum() <= 0.02 * len(cos) + 8
    return drc, dec


@pytest.mark.parametrize("n,open_boundary,normals,uvs,kw", [(3, False, True, True, {}), (12, False, True, True, {}), (40, False, True, True, {}), (33, True, True, True, {}),
                                                            (64, False, False, False, {}), (90, True, False, True, {}), (50, False, True, False, {}),
                                                            (30, False, True, True, dict(pos_bits=14, uv_bits=12)), (25, True, True, True, dict(pos_bits=20, uv_bits=16))])
def test_whole_file_round_trip_on_grids(n, open_boundary, normals, uvs, kw):
    mesh = synth.torus_mesh(n, normals=normals, uvs=uvs, open_boundary=open_boundary)
    _check_mesh(mesh, dmi.Config(**kw), kw.get("pos_bits", 11), kw.get("uv_bits", 10))


def _punched(n, frac, seed, open_boundary, normals):
    faces, pos, nrm, uv = synth.torus_grid(n, open_boundary=open_boundary)
    rng = np.random.default_rng(seed)
    keep = rng.random(len(faces)) > frac
    corner = faces[keep].ravel()
    cuv = uv[corner].copy()
    cuv[np.repeat((np.arange(keep.sum()) % 5) == 0, 3)] += np.float32(0.25)   # UV seams around every fifth face
    b = dmi.MeshBuilder()
    pid = b.add_attribute(pos[corner], dmi.ATT_POSITION)
    if normals:
        b.add_attribute(nrm[corner], dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, parents=[pid])
    b.add_attribute(cuv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, parents=[pid])
    b.set_connectivity_attribute(np.arange(len(corner), dtype=np.uint32).reshape(-1, 3))
    return b.build()


@pytest.mark.parametrize("n,frac,seed,open_boundary,normals", [(20, 0.03, 1, True, True), (30, 0.1, 2, True, False), (40, 0.02, 3, False, True), (25, 0.3, 4, False, True), (60, 0.05, 5, True, True)])
def test_whole_file_round_trip_with_holes_handles_seams_and_components(n, frac, seed, open_boundary, normals):
    """Faces knocked out at random: many boundary loops (topology splits), handles, components whose traversal starts at a boundary or
    inside, UV seams → the texture coordinates' own corner table."""
    _check_mesh(_punched(n, frac, seed, open_boundary, normals))


@pytest.mark.parametrize("name", ["tetrahedron", "cube_quads", "sphere", "punctured_sphere", "torus"])
def test_whole_file_round_trip_on_fixtures(name):
    from helpers import obj_session, product_mesh_from_oracle
    _check_mesh(product_mesh_from_oracle(obj_session(name)))


def test_whole_file_round_trip_custom_and_colour_attributes():
    rng = np.random.default_rng(11)
    faces, pos, nrm, uv = synth.torus_grid(24)
    corner = faces.ravel()
    b = dmi.MeshBuilder()
    pid = b.add_attribute(pos[corner], dmi.ATT_POSITION)
    b.add_attribute(nrm[corner], dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, parents=[pid])
    b.add_attribute((np.arange(len(corner)) // 30).astype(np.uint32).reshape(-1, 1), dmi.ATT_CUSTOM, dmi.DOMAIN_CORNER)
    b.add_attribute(rng.uniform(0, 1, size=(len(corner), 4)).astype(np.float32), dmi.ATT_COLOR, dmi.DOMAIN_CORNER)
    b.set_connectivity_attribute(np.arange(len(corner), dtype=np.uint32).reshape(-1, 3))
    _check_mesh(b.build())


def test_one_million_triangles_whole_file():
    mesh = synth.torus_mesh(708)
    drc = dmi.encode_mesh(mesh)
    dec = dmi.decode_mesh(drc)
    assert dec["faces"].shape == (len(mesh.faces), 3) and dec["num_points"] == mesh.attributes[0].num_points
    q, mn, rng = numpy_quantize(mesh.attributes[0].values, 11)
    in_faces = np.asarray(mesh.faces, np.int64).reshape(-1, 3)
    got = _requantize(dec["attributes"][0]["values"], mn, rng, 11)[dec["faces"].astype(np.int64)]
    assert (_canonical_faces(q[in_faces]) == _canonical_faces(got)).all()


def test_malformed_files_are_error_codes():
    mesh = synth.torus_mesh(8)
    good = dmi.encode_mesh(mesh)
    for cut in (0, 4, 11, 30, len(good) // 2, len(good) - 1):
        with pytest.raises(dmi.DracoMiError):
            dmi.decode_mesh(good[:cut])
    with pytest.raises(dmi.DracoMiError):
        dmi.decode_mesh(b"DRACX" + good[5:])
    rng = np.random.default_rng(9)
    for _ in range(200):   # flipped bits: an error code or some other mesh, never a crash
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(5, len(b)))] ^= 1 << int(rng.integers(0, 8))
        try:
            dmi.decode_mesh(bytes(b))
        except dmi.DracoMiError:
            pass
