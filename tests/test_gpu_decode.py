"""Round trip through the oracle's decoder (oracle/orc_decode.cpp) of what the PRODUCT wrote: the attribute section from the device
decodes — with the bytes and the connectivity stage alone — to the quantized values an independent numpy quantizer computes from the
inputs (a third implementation of quantization_coordinate_wise.rs:70-91, f32 operation by f32 operation), and dequantizes to within
half a step of the inputs; normals come back within the 8-bit octahedral grid's error."""
import numpy as np
import pytest

import draco_oxide_amd as dmi
import orc
from draco_oxide_amd import synth
from helpers import oracle_from_product_mesh

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
