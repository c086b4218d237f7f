"""Accessors copied up where they lie (round 5; dmi_host_alloc + dmi_build.cpp "in place"): a primitive whose arrays sit in page-locked memory of the
library's goes up by DMA as it is (rows keep their stride, u8 / u16 indices are widened on the device) instead of being packed into staging by host
threads — the built mesh is the host builder's either way."""
import numpy as np
import pytest

import draco_oxide_amd as dmi
from draco_oxide_amd import binding, gltf, synth

pytestmark = pytest.mark.gpu


def _same_mesh(a, b):
    assert a.faces.shape == b.faces.shape and (a.faces == b.faces).all()
    assert len(a.attributes) == len(b.attributes)
    for x, y in zip(a.attributes, b.attributes):
        assert x.values.tobytes() == y.values.tobytes()
        assert (x.point_to_value is None) == (y.point_to_value is None)
        assert x.point_to_value is None or (x.point_to_value == y.point_to_value).all()
        assert (x.num_points, x.unique_id, x.att_type, x.domain, x.parent_index) == (y.num_points, y.unique_id, y.att_type, y.domain, y.parent_index)


def _interleaved_in(buf, at, pos, nrm, uv, faces, index_dtype):
    """pos | nrm | uv as 32-byte records + the indices, written into the uint8 array `buf` from offset `at` (4-byte aligned) → (RawMesh of VIEWS, end)."""
    n = len(pos)
    rec = buf[at: at + 32 * n].view(np.float32).reshape(n, 8)
    rec[:, 0:3], rec[:, 3:6], rec[:, 6:8] = pos, nrm, uv
    at += 32 * n
    idx = buf[at: at + faces.size * np.dtype(index_dtype).itemsize].view(index_dtype)
    idx[:] = faces.ravel()
    at = (at + idx.nbytes + 3) & ~3
    rm = dmi.RawMesh()
    rm.add_attribute(rec[:, 0:3], dmi.ATT_POSITION, dmi.DOMAIN_POSITION, [])
    rm.add_attribute(rec[:, 3:6], dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, [0])
    rm.add_attribute(rec[:, 6:8], dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, [0])
    rm.set_indices(idx)
    return rm, at


def _host_built(pos, nrm, uv, faces):
    b = dmi.MeshBuilder()
    pid = b.add_attribute(pos, dmi.ATT_POSITION)
    b.add_attribute(nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, parents=[pid])
    b.add_attribute(uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, parents=[pid])
    b.set_connectivity_attribute(faces)
    return b.build()


def test_rows_and_indices_gathered_out_of_page_locked_memory():
    hb = binding.HostBuffer(8 << 20)
    try:
        raws, wants, at = [], [], 64
        for n, dt, soup in [(12, np.uint8, False), (40, np.uint16, False), (90, np.uint32, False), (30, np.uint16, True)]:
            faces, pos, nrm, uv = synth.torus_grid(n, seed=100 + n)
            if soup:   # every corner its own point: value dedup + point merge on rows the device read in place
                c = faces.ravel()
                pos, nrm, uv, faces = pos[c], nrm[c], uv[c], np.arange(len(c), dtype=np.uint32).reshape(-1, 3)
            rm, at = _interleaved_in(hb.array, at, pos, nrm, uv, faces, dt)
            raws.append(rm)
            wants.append(_host_built(pos, nrm, uv, faces))
        with dmi.meshes_build(raws, host_values=True) as batch:
            tm = dmi.last_build_timings()
            assert tm["device_meshes"] == 4 and tm["in_place_meshes"] == 4 and tm["pack_ms"] < 1.0
            for j, w in enumerate(wants):
                _same_mesh(batch.mesh(j), w)
        # the same arrays plus one primitive in ordinary memory: two groups, one gathered, one packed — same meshes
        faces, pos, nrm, uv = synth.torus_grid(25, seed=7)
        plain = np.zeros(1 << 20, np.uint8)
        rm, _ = _interleaved_in(plain, 0, pos, nrm, uv, faces, np.uint16)
        with dmi.meshes_build(raws + [rm], host_values=True) as batch:
            tm = dmi.last_build_timings()
            assert tm["device_meshes"] == 5 and tm["in_place_meshes"] == 4
            for j, w in enumerate(wants + [_host_built(pos, nrm, uv, faces)]):
                _same_mesh(batch.mesh(j), w)
    finally:
        hb.free()


def test_files_read_into_library_memory_transcode_to_the_same_files(monkeypatch):
    """An importer that reads its files into dmi_host_alloc memory (binding.HostBuffer): the transcode copies their accessors up in place
    (stats: every buffer in place), and the output files are those of the same bytes in ordinary memory, packed and copied."""
    glbs, _ = synth.batch_glbs(24, lo=500, hi=30000, seed=77)
    plain, st0 = binding.transcode_assets(glbs)
    assert st0["buffers_in_place"] == 0 and st0["primitives_in_place"] == 0 and st0["primitives_device_built"] == 24
    held = [binding.HostBuffer.holding(g) for g in glbs]
    try:
        assert binding.load_library().dmi_host_is_registered(held[0].array.ctypes.data + 64, 16) == 1
        placed, st1 = binding.transcode_assets([h.view() for h in held])
        assert st1["buffers_in_place"] == 24 and st1["primitives"] == 24 and st1["primitives_in_place"] == 24 and st1["primitives_host_built"] == 0
        assert [bytes(g) for g, _ in placed] == [bytes(g) for g, _ in plain]
        monkeypatch.setenv("DMI_NO_IN_PLACE", "1")                      # (A/B switch: pack even what could go up in place)
        packed, _ = binding.transcode_assets([h.view() for h in held])
        assert [bytes(g) for g, _ in packed] == [bytes(g) for g, _ in plain]
    finally:
        addr = held[0].array.ctypes.data
        for h in held:
            h.free()
    assert binding.load_library().dmi_host_is_registered(addr + 64, 16) == 0      # parked blocks are not "in place"
    again = binding.HostBuffer(len(glbs[0]))                       # … and are handed out again
    again.free()
