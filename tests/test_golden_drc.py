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
