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
