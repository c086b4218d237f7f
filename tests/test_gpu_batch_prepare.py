"""dmi_meshes_prepare, device form (csrc/dmi_prepare.cpp prepare_slice_device + csrc/dmi_conn.hip): the universal corner tables of all
meshes of a batch from one launch per kernel, the host walks per mesh on threads, then the coding-order relabelling / fan rows / map
compositions of all jobs in one launch per kernel.  Every mesh of a mixed batch must give the oracle's `.drc` — the same bytes as the
per-mesh host form (DMI_HOST_CONNECTIVITY=1) — whatever path it takes: seam-free grids (deferred jobs), meshes built by MeshBuilder (point →
value maps on every attribute), UV seams (an attribute table of its own → host relabelling), soups (device flags → the reference's
serial walks), tiny meshes."""
import numpy as np
import pytest

import draco_oxide_amd as dmi
import orc
from draco_oxide_amd import synth
from helpers import obj_session, oracle_from_product_mesh, product_mesh_from_oracle
from test_gpu_parity import _assert_same, _cones, _heavy_tailed_mesh, _soup_mesh

pytestmark = pytest.mark.gpu


def _seam_mesh(n=20):
    faces, pos, nrm, uv = synth.torus_grid(n)
    cp = faces.ravel()
    cuv = uv[cp].copy()
    cuv[np.repeat((np.arange(len(faces)) % 5) == 0, 3)] += np.float32(0.25)
    b = dmi.MeshBuilder()
    pid = b.add_attribute(pos[cp], dmi.ATT_POSITION)
    b.add_attribute(nrm[cp], dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, parents=[pid])
    b.add_attribute(cuv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, parents=[pid])
    b.set_connectivity_attribute(np.arange(len(cp), dtype=np.uint32).reshape(-1, 3))
    return b.build()


def _mixed_batch():
    meshes = [synth.torus_mesh(n, normals=nr, uvs=uv, open_boundary=ob, seed=100 + n)
              for n, nr, uv, ob in [(8, True, True, False), (31, True, True, True), (64, False, False, False), (47, True, False, False), (90, False, True, True), (3, True, True, False)]]
    meshes += [product_mesh_from_oracle(obj_session(name)) for name in ("tetrahedron", "cube_quads", "sphere", "punctured_sphere", "torus")]
    meshes.append(_seam_mesh())
    meshes.append(_heavy_tailed_mesh(40, 3))
    for seed in (1, 2, 3):
        m, sess = _soup_mesh(seed, uv_per_corner=(seed % 2 == 0))
        try:
            sess.encode()
        except orc.OracleError:
            continue
        meshes.append(m)
    meshes.append(_cones(9, True, False, 4))
    meshes.append(_cones(13, False, True, 5))
    return meshes


def _encode_batch(meshes, cfg=None):
    jobs = dmi.meshes_prepare(meshes, cfg or dmi.Config.default())
    try:
        sections = dmi.jobs_encode(jobs)
        return [j.header_and_connectivity + s for j, s in zip(jobs, sections)]
    finally:
        for j in jobs:
            j.close()


def test_mixed_batch_device_form_equals_oracle_and_host_form(monkeypatch):
    meshes = _mixed_batch()
    want = [oracle_from_product_mesh(m).encode() for m in meshes]
    got = _encode_batch(meshes)
    for k, (g, w) in enumerate(zip(got, want)):
        _assert_same(g, w, f"mixed batch, device form, mesh {k} ({len(meshes[k].faces)} faces)")
    monkeypatch.setenv("DMI_HOST_CONNECTIVITY", "1")
    got = _encode_batch(meshes)
    for k, (g, w) in enumerate(zip(got, want)):
        _assert_same(g, w, f"mixed batch, host form, mesh {k}")


def test_batch_jobs_also_encode_one_by_one_and_repeatedly():
    meshes = synth.batch_meshes(12, lo=500, hi=20000, seed=9)
    jobs = dmi.meshes_prepare(meshes, dmi.Config.default())
    try:
        first = dmi.jobs_encode(jobs)
        again = dmi.jobs_encode(jobs)
        assert first == again
        for k, j in enumerate(jobs):
            single = j.encode()
            assert single == first[k], k
            _assert_same(j.header_and_connectivity + single, oracle_from_product_mesh(meshes[k]).encode(), f"batch job {k} encoded alone")
    finally:
        for j in jobs:
            j.close()


@pytest.mark.parametrize("bits", [(14, 12), (8, 7), (20, 16)])
def test_batch_device_form_at_other_bit_widths(bits):
    meshes = synth.batch_meshes(6, lo=300, hi=8000, seed=21)
    cfg = dmi.Config(pos_bits=bits[0], uv_bits=bits[1])
    got = _encode_batch(meshes, cfg)
    for k, m in enumerate(meshes):
        _assert_same(got[k], oracle_from_product_mesh(m).encode(pos_bits=bits[0], uv_bits=bits[1]), f"bits {bits}, mesh {k}")


def test_batch_with_an_invalid_mesh_fails_as_a_whole():
    good = synth.torus_mesh(10)
    pos = np.random.default_rng(1).random((4, 3), dtype=np.float32)
    unused = dmi.Mesh(np.asarray([[0, 1, 3]], np.uint32), [dmi.Attribute(pos, dmi.ATT_POSITION)])   # vertex 2 is never used: the reference panics
    with pytest.raises(dmi.DracoMiError):
        dmi.meshes_prepare([good, unused, good], dmi.Config.default())
    bad = dmi.Mesh(np.asarray([[0, 1, 9]], np.uint32), [dmi.Attribute(pos, dmi.ATT_POSITION)])
    with pytest.raises(dmi.DracoMiError):
        dmi.meshes_prepare([good, bad], dmi.Config.default())


def test_large_batch_matches_single_encodes():
    meshes = synth.batch_meshes(96, lo=2e3, hi=6e4, seed=5)
    got = _encode_batch(meshes)
    for k in range(0, len(meshes), 7):
        assert got[k] == dmi.encode_mesh(meshes[k]), k
    k = int(np.argmax([len(m.faces) for m in meshes]))
    _assert_same(got[k], oracle_from_product_mesh(meshes[k]).encode(), "largest mesh of the batch")
