"""The two places a single-job encode can run its serial coders: device chains (k_chains) and the hybrid form (symbols + device-built
tables back over PCIe, every stream on a host core: csrc/host_chains.cpp).  DMI_CHAINS=device|host (read at job creation) forces one;
the default picks by the longest stream.  Same bytes either way, and the oracle's."""
import numpy as np
import pytest

import draco_oxide_amd as dmi
import orc
from draco_oxide_amd import synth
from helpers import oracle_from_product_mesh
from test_gpu_parity import _assert_same, _cones, _heavy_tailed_mesh, _soup_mesh

pytestmark = pytest.mark.gpu


def _both_modes(mesh, want, what, monkeypatch, cfg=None):
    for mode in ("host", "device"):
        monkeypatch.setenv("DMI_CHAINS", mode)
        _assert_same(dmi.encode_mesh(mesh, cfg), want, f"{what} (chains: {mode})")
    monkeypatch.delenv("DMI_CHAINS")


@pytest.mark.parametrize("n,open_boundary,normals,uvs", [(5, False, True, True), (40, False, True, True), (33, True, True, True), (64, False, False, False),
                                                         (150, True, False, True), (90, False, True, False)])
def test_synthetic_grids_both_chain_forms(n, open_boundary, normals, uvs, monkeypatch):
    mesh = synth.torus_mesh(n, normals=normals, uvs=uvs, open_boundary=open_boundary)
    _both_modes(mesh, oracle_from_product_mesh(mesh).encode(), f"grid {n}", monkeypatch)


@pytest.mark.parametrize("seed", [1, 2, 4])
def test_soups_with_seams_both_chain_forms(seed, monkeypatch):
    mesh, sess = _soup_mesh(seed, uv_per_corner=(seed % 2 == 0))
    try:
        want = sess.encode()
    except orc.OracleError:
        pytest.skip("reference rejects this soup")
    _both_modes(mesh, want, f"soup {seed}", monkeypatch)


@pytest.mark.parametrize("n,pos_bits,uv_bits", [(48, 11, 10), (48, 16, 14), (25, 20, 16), (30, 1, 1)])
def test_heavy_tails_rare_symbols_both_chain_forms_and_table_forms(n, pos_bits, uv_bits, monkeypatch):
    """Sparse alphabets put frequency-1 and multi-byte-renormalisation symbols into the streams: the general step of the host coder.
    Also with the table stage on the host (DMI_HOST_TABLES=1): the host chains then take their tables from the host normaliser."""
    mesh = _heavy_tailed_mesh(n, seed=n * 31 + pos_bits)
    cfg = dmi.Config(pos_bits=pos_bits, uv_bits=uv_bits)
    try:
        want = oracle_from_product_mesh(mesh).encode(pos_bits=pos_bits, uv_bits=uv_bits)
    except orc.OracleError:
        pytest.skip("reference cannot code this table")
    _both_modes(mesh, want, f"heavy tails n={n}", monkeypatch, cfg)
    monkeypatch.setenv("DMI_HOST_TABLES", "1")
    _both_modes(mesh, want, f"heavy tails n={n}, host tables", monkeypatch, cfg)


def test_custom_attribute_and_cones_both_chain_forms(monkeypatch):
    faces, pos, nrm, uv = synth.torus_grid(14)
    corner = faces.ravel()
    b = dmi.MeshBuilder()
    pid = b.add_attribute(pos[corner], dmi.ATT_POSITION)
    b.add_attribute(uv[corner], dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, parents=[pid])
    feat = (np.arange(len(corner)) // 12).astype(np.uint32).reshape(-1, 1)
    b.add_attribute(feat, dmi.ATT_CUSTOM, dmi.DOMAIN_CORNER)
    f2 = np.arange(len(corner), dtype=np.uint32).reshape(-1, 3)
    b.set_connectivity_attribute(f2)
    want = orc.Session.from_arrays(f2, [dict(data=pos[corner], type=orc.POSITION), dict(data=uv[corner], type=orc.TEXCOORD, domain=orc.DOM_CORNER, parents=[0]),
                                        dict(data=feat, type=orc.CUSTOM, domain=orc.DOM_CORNER)]).encode()
    _both_modes(b.build(), want, "custom attribute (host-table form)", monkeypatch)
    m = _cones(13, False, True, seed=13)
    _both_modes(m, oracle_from_product_mesh(m).encode(), "cones", monkeypatch)


def test_default_picks_the_form_by_stream_length_and_reports_it():
    small = dmi.mesh_prepare(synth.torus_mesh(20), dmi.Config(flags=dmi.FLAG_TIMINGS))      # 1200 position symbols
    large = dmi.mesh_prepare(synth.torus_mesh(128), dmi.Config(flags=dmi.FLAG_TIMINGS))     # 49152 position symbols
    small.encode(); large.encode()
    assert small.timings()["host_chains"] == 0 and large.timings()["host_chains"] == 1
    t = large.timings()
    assert t["num_streams"] == 5 and t["longest_stream_ms"] > 0 and t["rans_ms"] >= t["longest_stream_ms"]
    small.close(); large.close()


def test_hybrid_job_reuse_and_zero_normal_error(monkeypatch):
    monkeypatch.setenv("DMI_CHAINS", "host")
    mesh = synth.torus_mesh(60)
    job = dmi.mesh_prepare(mesh)
    a, b = job.encode(), job.encode()
    assert a == b
    _assert_same(job.header_and_connectivity + a, oracle_from_product_mesh(mesh).encode(), "hybrid job reuse")
    job.close()
    faces, pos, nrm, _ = synth.torus_grid(10)
    nrm = nrm.copy()
    nrm[7] = 0
    bad = dmi.Mesh(faces, [dmi.Attribute(pos, dmi.ATT_POSITION), dmi.Attribute(nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, unique_id=1, parent_index=0)])
    with pytest.raises(dmi.DracoMiError) as e:
        dmi.encode_mesh(bad)
    assert e.value.status == 6


def test_one_million_triangles_on_device_chains(monkeypatch):
    """Large single meshes take the hybrid form by default; the device chain stays covered at size (positions-only 1M, parallelogram)."""
    monkeypatch.setenv("DMI_CHAINS", "device")
    mesh = synth.torus_mesh(707, normals=False, uvs=False)
    _assert_same(dmi.encode_mesh(mesh), oracle_from_product_mesh(mesh).encode(), "1M positions, device chains")


@pytest.mark.parametrize("toggle", ["DMI_NO_PACKED", "DMI_NO_SYM16"])
def test_packed_values_and_16_bit_symbols_are_layouts_not_arithmetic(toggle, monkeypatch):
    """The attributes of a fused sweep keep their quantized values packed (positions ≤ 21 bits in a uint64, normals in a uint16,
    texture coordinates ≤ 16 bits in a uint32) and symbols are uint16 where the alphabet allows; DMI_NO_PACKED / DMI_NO_SYM16 (read
    at job creation) switch the layouts off.  Same bytes — and the widths at which packing does not apply take the plain layout."""
    cases = [(synth.torus_mesh(40), None), (synth.torus_mesh(33, open_boundary=True), None), (synth.torus_mesh(64, uvs=False), None),
             (synth.torus_mesh(50, normals=False), None), (synth.torus_mesh(30), dmi.Config(pos_bits=15, uv_bits=15)),
             (synth.torus_mesh(30), dmi.Config(pos_bits=16, uv_bits=16)), (synth.torus_mesh(30), dmi.Config(pos_bits=21, uv_bits=12)),
             (synth.torus_mesh(30), dmi.Config(pos_bits=22, uv_bits=17)), (_cones(13, True, False, seed=3), None)]
    for mesh, cfg in cases:
        kw = {} if cfg is None else dict(pos_bits=cfg.pos_bits, uv_bits=cfg.uv_bits)
        try:
            want = oracle_from_product_mesh(mesh).encode(**kw)
        except orc.OracleError:
            continue
        _assert_same(dmi.encode_mesh(mesh, cfg), want, f"packed layouts {kw}")
        monkeypatch.setenv(toggle, "1")
        _assert_same(dmi.encode_mesh(mesh, cfg), want, f"{toggle} {kw}")
        monkeypatch.delenv(toggle)


def test_lds_window_sweep_gives_the_same_bytes(monkeypatch):
    """DMI_FUSED_WINDOWS=1: the packed fused sweep with its prediction neighbourhoods staged through LDS (k_predict_window_*; slower
    than the plain sweep on MI355X, kept as the recorded experiment).  Closed / open grids, a mesh without UVs, one without normals,
    high-valence fans (row overflow → corner-table walk), ranks outside every window (two far-apart components)."""
    monkeypatch.setenv("DMI_FUSED_WINDOWS", "1")
    cases = [synth.torus_mesh(40), synth.torus_mesh(150, open_boundary=True), synth.torus_mesh(64, uvs=False), synth.torus_mesh(50, normals=False), _cones(13, True, False, seed=3)]
    f1, p1, n1, u1 = synth.torus_grid(30, seed=1)
    f2, p2, n2, u2 = synth.torus_grid(41, seed=2, open_boundary=True)
    two = dmi.Mesh(np.concatenate([f1, f2 + np.uint32(len(p1))]), [dmi.Attribute(np.concatenate([p1, p2 + np.float32(4.0)]), dmi.ATT_POSITION),
                                                                   dmi.Attribute(np.concatenate([n1, n2]), dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, 1, 0),
                                                                   dmi.Attribute(np.concatenate([u1, u2]), dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, 2, 0)])
    cases.append(two)
    for mesh in cases:
        _assert_same(dmi.encode_mesh(mesh), oracle_from_product_mesh(mesh).encode(), "LDS-window sweep")
