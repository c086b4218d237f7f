"""Committed `.drc` goldens of the reference's OBJ fixtures (tests/golden/manifest.json, written by scripts/make_golden_drc.py).
source "restatement": bytes of the CPU oracle at the time of the commit — a regression anchor for oracle and product alike.
tests/golden/reference_drc/<name>.drc (absent here: the Rust crate cannot be built in this image) takes precedence when present:
files written by the reference's own tests/compatibility.rs pin the bytes to the real encoder."""
import hashlib
import json
import os

import pytest

from helpers import obj_session, product_mesh_from_oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))


def _expected(name):
    ref = os.path.join(GOLDEN, "reference_drc", name + ".drc")
    if os.path.exists(ref):
        return open(ref, "rb").read(), "reference"
    blob = open(os.path.join(GOLDEN, MANIFEST[name]["file"]), "rb").read()
    assert hashlib.sha256(blob).hexdigest() == MANIFEST[name]["sha256"] and len(blob) == MANIFEST[name]["bytes"]
    return blob, MANIFEST[name]["source"]


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_oracle_reproduces_the_golden_drc(name):
    want, source = _expected(name)
    got = obj_session(name).encode(dump=False)
    assert got == want, f"{name}: oracle output differs from the {source} golden ({len(got)} vs {len(want)} bytes)"


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_library_reproduces_the_golden_drc(name):
    import draco_oxide_amd as dmi
    want, source = _expected(name)
    sess = obj_session(name)
    sess.encode()
    got = dmi.encode_mesh(product_mesh_from_oracle(sess))
    assert got == want, f"{name}: library output differs from the {source} golden ({len(got)} vs {len(want)} bytes)"


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_golden_connectivity_decodes_back_to_the_fixture(name):
    """The committed file's connectivity section, decoded from its bytes alone, is the fixture's mesh (host only)."""
    import numpy as np
    import draco_oxide_amd as dmi
    want, _ = _expected(name)
    mesh = product_mesh_from_oracle(obj_session(name))
    dec = dmi.decode_connectivity(want)
    t = dec["tables"][0]
    assert t["num_faces"] == len(mesh.faces) and len(dec["seeds"]) == len(mesh.faces)
    conn = dmi.encode_connectivity(mesh)
    assert dec["consumed"] == len(conn.bytes) and t["num_vertices"] == conn.table(0)["num_vertices"]
    conn.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_golden_drc_decodes_back_to_the_fixture(name):
    """dmi_decode_mesh on the committed bytes: the triangles of the OBJ fixture, labelling-free, at the file's quantization."""
    import numpy as np
    import draco_oxide_amd as dmi
    from test_gpu_decode import numpy_quantize
    from test_gpu_decode_mesh import _canonical_faces, _requantize
    want, _ = _expected(name)
    mesh = product_mesh_from_oracle(obj_session(name))
    dec = dmi.decode_mesh(want)
    in_faces = np.asarray(mesh.faces, np.int64).reshape(-1, 3)
    assert dec["faces"].shape == in_faces.shape and len(dec["attributes"]) == len(mesh.attributes)
    rows_in, rows_out = [], []
    for att, d in zip(mesh.attributes, dec["attributes"]):
        if d["portabilization"] != 2:
            continue
        q, mn, rng = numpy_quantize(att.values, d["bits"])
        q = q if att.point_to_value is None else q[att.point_to_value]
        rows_in.append(q[in_faces])
        rows_out.append(_requantize(d["values"], mn, rng, d["bits"])[dec["faces"].astype(np.int64)])
    assert (_canonical_faces(np.concatenate(rows_in, axis=2)) == _canonical_faces(np.concatenate(rows_out, axis=2))).all()
