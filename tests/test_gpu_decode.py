"""Round trip through the oracle's decoder (oracle/orc_decode.cpp) of what the PRODUCT wrote: the attribute section from the device
decodes — with the bytes and the connectivity stage alone — to the quantized values an independent numpy quantizer computes from the
inputs (a third implementation of quantization_coordinate_wise.rs:70-91, f32 operation by f32 operation), and dequantizes to within
half a step of the inputs; normals come back within the 8-bit octahedral grid's error."""
import numpy as np
import pytest

import draco_oxide_amd as dmi
import orc
from draco_oxide_amd import synth
from helpers import oracle_values_by_point, oracle_from_product_mesh

pytestmark = pytest.mark.gpu


def numpy_quantize(values, bits):
    """quantization_coordinate_wise.rs:24-91 in numpy float32: min/max seeded with 0.0, ONE range for all components, every operation
    rounded to f32, `as i64 as i32` truncation."""
    v = np.asarray(values, np.float32)
    mn = np.minimum(v.min(axis=0), np.float32(0.0)).astype(np.float32)
    mx = np.maximum(v.max(axis=0), np.float32(0.0)).astype(np.float32)
    rng = np.float32((mx - mn).astype(np.float32).max())
    maxq = np.float32((1 << bits) - 1)
    diff = (v - mn).astype(np.float32)
    norm = diff if rng == 0 else (diff / rng).astype(np.float32)
    q = ((norm * maxq).astype(np.float32) + np.float32(0.5)).astype(np.float32)
    return np.trunc(q).astype(np.int64).astype(np.int32), mn, rng


@pytest.mark.parametrize("n,open_boundary,kw", [(40, False, {}), (33, True, {}), (128, False, {}), (60, False, dict(pos_bits=14, uv_bits=12)), (50, True, dict(pos_bits=20, uv_bits=16))])
def test_product_section_decodes_to_the_numpy_quantization_of_the_inputs(n, open_boundary, kw):
    mesh = synth.torus_mesh(n, open_boundary=open_boundary)
    job = dmi.mesh_prepare(mesh, dmi.Config(**kw))
    section = job.encode()
    job.close()
    sess = oracle_from_product_mesh(mesh)
    decoded, used = sess.decode_attributes(section)
    assert used == len(section) and len(decoded) == 3
    pos, nrm, uv = [a.values for a in mesh.attributes]
    for d, vals, bits in ((decoded[0], pos, kw.get("pos_bits", 11)), (decoded[2], uv, kw.get("uv_bits", 10))):
        q, mn, rng = numpy_quantize(vals, bits)
        assert (d["portable"] == q[d["points"]]).all()
        step = float(rng) / ((1 << bits) - 1)
        assert np.abs(d["values"] - vals[d["points"]]).max() <= 0.5001 * step + 1e-6
    cos = (nrm[decoded[1]["points"]] * decoded[1]["values"]).sum(axis=1)
    assert cos.min() > np.cos(np.radians(2.5))


def test_product_batch_sections_decode():
    meshes = [synth.torus_mesh(10 + 7 * k, seed=50 + k, open_boundary=bool(k & 1)) for k in range(6)]
    jobs = dmi.meshes_prepare(meshes)
    outs = dmi.jobs_encode(jobs)
    for m, o in zip(meshes, outs):
        decoded, used = oracle_from_product_mesh(m).decode_attributes(o)
        assert used == len(o)
        q, _, _ = numpy_quantize(m.attributes[0].values, 11)
        assert (decoded[0]["portable"] == q[decoded[0]["points"]]).all()
    for j in jobs:
        j.close()


# ---- dmi_decode_attributes: the product's own decoder-side path (host cores for the serial stages, device for normals + dequantization) ----
def _decode_with_product(mesh, section, want_tables=False):
    conn = dmi.encode_connectivity(mesh)
    tables = [conn.table(i) for i in range(conn.num_tables)]
    seeds = conn.seeds()
    conn.close()
    got = dmi.decode_attributes(section, tables, mesh.attributes[0].num_points, seeds=seeds)
    return (got, tables) if want_tables else got


def _expand(att):
    """An attribute's values per point."""
    return att.values if att.point_to_value is None else att.values[att.point_to_value]


def _check_against_oracle_decoder(mesh, section, kw=None):
    """The product's decoder and the oracle's decoder agree value for value (per point), and both sit within the quantization error
    of the inputs."""
    kw = kw or {}
    got, tables = _decode_with_product(mesh, section, want_tables=True)
    sess = oracle_from_product_mesh(mesh)
    ref, used = sess.decode_attributes(section)
    assert used == len(section) and len(got) == len(ref) == len(mesh.attributes)
    for i, (g, d, att) in enumerate(zip(got, ref, mesh.attributes)):
        assert g["att_type"] == att.att_type and g["num_components"] == att.values.shape[1] and g["unique_id"] == att.unique_id
        per_point, seen = oracle_values_by_point(tables[i], tables[0], d, len(g["values"]))   # every point a corner references
        if g["portabilization"] == 3:
            assert np.abs(g["values"][seen] - per_point[seen]).max() < 2e-6, f"attribute {i}: normals differ between the two decoders"
            n = _expand(att)
            n = n / np.linalg.norm(n, axis=1, keepdims=True)
            cos = (n[seen] * g["values"][seen]).sum(axis=1)
            # (the reference's non-injective diamond inversion, DESIGN §2: normals with a zero octahedral coordinate behind a prediction outside
            #  the diamond collapse onto another axis point — in BOTH decoders, checked above; they are left out of the angle check)
            generic = (np.abs(n[seen]) > 1e-3).all(axis=1)
            lossy = (cos < np.cos(np.radians(2.5))) & generic
            assert lossy.sum() <= 0.02 * len(cos) + 8, f"attribute {i}: {lossy.sum()} normals off"
        else:
            assert (g["values"][seen] == per_point[seen]).all(), f"attribute {i}: values differ between the two decoders"
            if g["portabilization"] == 2:
                raw = _expand(att)
                bits = g["bits"]
                rng = float((np.maximum(att.values.max(axis=0), 0.0) - np.minimum(att.values.min(axis=0), 0.0)).max())
                step = rng / ((1 << bits) - 1) if rng > 0 else 0.0
                assert np.abs(g["values"][seen] - raw[seen]).max() <= 0.5001 * step + 1e-6 * max(rng, 1.0)
            else:
                assert (g["values"][seen] == _expand(att)[seen].view(np.uint32)).all()
    return got


@pytest.mark.parametrize("n,open_boundary,normals,uvs,kw", [(12, False, True, True, {}), (40, False, True, True, {}), (33, True, True, True, {}), (64, False, False, False, {}),
                                                            (90, True, False, True, {}), (50, False, True, False, {}), (30, False, True, True, dict(pos_bits=14, uv_bits=12)),
                                                            (25, False, True, True, dict(pos_bits=20, uv_bits=16))])
def test_product_decoder_round_trip_on_grids(n, open_boundary, normals, uvs, kw):
    mesh = synth.torus_mesh(n, normals=normals, uvs=uvs, open_boundary=open_boundary)
    job = dmi.mesh_prepare(mesh, dmi.Config(**kw))
    section = job.encode()
    job.close()
    _check_against_oracle_decoder(mesh, section, kw)


def test_product_decoder_round_trip_seams_custom_and_soups():
    """Attribute tables with seams (lone-normal fan walks on the normal's own table), point_to_value maps, a Custom (ToBits) attribute,
    a generic colour attribute (delta + difference), non-manifold soups."""
    from test_gpu_parity import _soup_mesh
    rng = np.random.default_rng(11)
    faces, pos, nrm, uv = synth.torus_grid(24)
    corner = faces.ravel()
    cuv = uv[corner].copy()
    cuv[np.repeat((np.arange(len(faces)) % 7) == 0, 3)] += np.float32(0.5)
    cn = nrm[corner].copy()
    cn[::5] = cn[0]
    b = dmi.MeshBuilder()
    pid = b.add_attribute(pos[corner], dmi.ATT_POSITION)
    b.add_attribute(cn, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, parents=[pid])
    b.add_attribute(cuv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, parents=[pid])
    b.add_attribute((np.arange(len(corner)) // 30).astype(np.uint32).reshape(-1, 1), dmi.ATT_CUSTOM, dmi.DOMAIN_CORNER)
    b.add_attribute(rng.uniform(0, 1, size=(len(corner), 4)).astype(np.float32), dmi.ATT_COLOR, dmi.DOMAIN_CORNER)
    b.set_connectivity_attribute(np.arange(len(corner), dtype=np.uint32).reshape(-1, 3))
    meshes = [b.build()]
    for seed in (2, 3, 4):
        m, sess = _soup_mesh(seed, uv_per_corner=(seed % 2 == 0))
        try:
            sess.encode()
            meshes.append(m)
        except orc.OracleError:
            pass
    for mesh in meshes:
        job = dmi.mesh_prepare(mesh)
        section = job.encode()
        job.close()
        _check_against_oracle_decoder(mesh, section)


def test_product_decoder_at_one_million_triangles_and_on_garbage():
    mesh = synth.torus_mesh(707)
    job = dmi.mesh_prepare(mesh)
    section = job.encode()
    job.close()
    got = _decode_with_product(mesh, section)
    pos = mesh.attributes[0].values
    step = float((np.maximum(pos.max(axis=0), 0) - np.minimum(pos.min(axis=0), 0)).max()) / 2047
    assert np.abs(got[0]["values"] - pos).max() <= 0.5001 * step + 1e-6
    cos = (mesh.attributes[1].values * got[1]["values"]).sum(axis=1)
    assert np.quantile(cos, 0.001) > np.cos(np.radians(2.5))
    conn = dmi.encode_connectivity(mesh)
    tables = [conn.table(i) for i in range(conn.num_tables)]
    with pytest.raises(dmi.DracoMiError):                       # a truncated section is an error code
        dmi.decode_attributes(section[: len(section) // 3], tables, mesh.attributes[0].num_points, seeds=conn.seeds())
    bad = bytearray(section)
    bad[40:44] = b"\xff\xff\xff\xff"
    try:                                                       # flipped bytes: garbage out or an error code, never a crash
        dmi.decode_attributes(bytes(bad), tables, mesh.attributes[0].num_points, seeds=conn.seeds())
    except dmi.DracoMiError:
        pass
    conn.close()


def test_product_decoder_survives_mutated_sections():
    """300 sections with flipped / truncated / spliced bytes: an error code or garbage values, never a crash or a hang."""
    mesh = synth.torus_mesh(16)
    job = dmi.mesh_prepare(mesh)
    section = job.encode()
    job.close()
    conn = dmi.encode_connectivity(mesh)
    tables = [conn.table(i) for i in range(conn.num_tables)]
    seeds = conn.seeds()
    conn.close()
    rng = np.random.default_rng(5)
    outcomes = [0, 0]
    for it in range(300):
        b = bytearray(section)
        kind = it % 3
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif kind == 1:
            b = b[: int(rng.integers(0, len(b)))]
        else:
            at = int(rng.integers(0, len(b)))
            b[at:at] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 9)), dtype=np.uint8))
        try:
            dmi.decode_attributes(bytes(b), tables, mesh.attributes[0].num_points, seeds=seeds)
            outcomes[0] += 1
        except dmi.DracoMiError:
            outcomes[1] += 1
    assert outcomes[1] > 50, outcomes


def test_product_decoder_with_vertices_no_corner_names():
    """The entropy decoders start beside the traversals with the table's vertex count as their entry count; a table that carries a vertex no
    corner names (the caller's tables may) has fewer entries, and those attributes are decoded again with the traversal's count: same values."""
    mesh = synth.torus_mesh(24)
    job = dmi.mesh_prepare(mesh)
    section = job.encode()
    job.close()
    want, tables = _decode_with_product(mesh, section, want_tables=True)
    conn = dmi.encode_connectivity(mesh)
    seeds = conn.seeds()
    conn.close()
    padded = []
    for t in tables:
        t = dict(t)
        t["sequence"] = None   # (let the decoder traverse)
        t["num_vertices"] = t["num_vertices"] + 3
        t["left_most_corner"] = np.concatenate([t["left_most_corner"], np.zeros(3, np.uint32)])
        padded.append(t)
    got = dmi.decode_attributes(section, padded, mesh.attributes[0].num_points, seeds=seeds)
    for g, w in zip(got, want):
        assert (g["values"].view(np.uint32) == w["values"].view(np.uint32)).all()
