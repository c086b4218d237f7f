"""BASELINE.json configs[3] and configs[4] at their stated sizes, under `-m gpu` (configs[1] and configs[2] at size live in
test_gpu_parity.py: test_one_million_triangles_positions_only_delta_config1, test_ten_million_triangles_full_attribute_set)."""
import os

import numpy as np
import pytest

import draco_oxide_amd as dmi
import orc
from draco_oxide_amd import synth
from helpers import oracle_from_product_mesh
from test_gpu_parity import _assert_same, _leb

pytestmark = pytest.mark.gpu


def check_attribute_section(a, n_vertices, limit=1 << 30):
    """Size-independent property of a pos+nrm+uv attribute section: every attribute's rANS stream decodes with the oracle's inverse
    coder to exactly V·N symbols and the section is consumed to the last byte.  Returns the decoded symbol arrays."""
    nA = a[0]
    assert nA == 3
    p = 1 + 3 * nA + 7 * nA
    counts = [n_vertices * 3, n_vertices * 2, n_vertices * 2]
    out = []
    for i in range(nA):
        scheme, transform, rans = a[p], a[p + 1], a[p + 2]
        assert rans == 1 and (scheme, transform) == [(1, 1), (6, 3), (5, 1)][i]
        p += 3
        syms, used = orc.decode_symbols(a[p:p + limit], counts[i])
        assert len(syms) == counts[i]
        out.append(syms)
        p += used
        if scheme == 6:      # transform meta (8), zero_prob (1), leb len + rABS bytes, oct bits (1)
            p += 9
            ln, p = _leb(a, p)
            p += ln + 1
        elif scheme == 5:    # u32 count, zero_prob, leb len + rABS bytes, transform meta (8), port meta (2*4+4+1)
            p += 5
            ln, p = _leb(a, p)
            p += ln + 8 + 13
        else:                # transform meta (8) + port meta (3*4+4+1)
            p += 8 + 17
    assert p == len(a)
    return out


def test_config3_batch_of_1024_meshes_one_call():
    """configs[3] shape on one GPU: 1024 independent meshes (F log-uniform in [2k, 200k], ≈45M triangles, pos+nrm+uv) prepared by
    dmi_meshes_prepare and coded by ONE dmi_jobs_encode.  Every one of the 3072 rANS streams decodes to V·N symbols with the section
    consumed exactly; a seeded sample of 64 meshes including the largest is compared byte for byte with the oracle."""
    n = int(os.environ.get("DMI_BATCH_N", "1024"))
    meshes = synth.batch_meshes(n)
    jobs = dmi.meshes_prepare(meshes)
    outs = dmi.jobs_encode(jobs)
    assert len(outs) == n
    for m, o in zip(meshes, outs):
        check_attribute_section(o, len(m.attributes[0].values), limit=len(o))
    sizes = [len(m.faces) for m in meshes]
    rng = np.random.default_rng(33)
    sample = sorted(set([int(np.argmax(sizes)), int(np.argmin(sizes)), 0, n - 1] + [int(k) for k in rng.choice(n, size=min(n, 60), replace=False)]))
    assert len(sample) >= min(n, 60)
    for k in sample:
        want = oracle_from_product_mesh(meshes[k]).encode(dump=False)
        _assert_same(jobs[k].header_and_connectivity + outs[k], want, f"batch mesh {k} ({sizes[k]} triangles)")
    # the single-job path (hybrid chains for the larger ones) gives the same section as the batch (device chains)
    for k in sample[:6]:
        assert jobs[k].encode() == outs[k]
    for j in jobs:
        j.close()


def _host_memory_gib():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / (1 << 20)
    except OSError:
        pass
    return 0.0


@pytest.mark.skipif(os.environ.get("DMI_SKIP_100M") == "1", reason="DMI_SKIP_100M=1")
def test_config4_hundred_million_triangles_14_bit_single_mesh():
    """configs[4] size on ONE GPU: a single 99 998 082-triangle mesh (n=7071), pos+nrm+uv, 14-bit positions — 33 GB resident.
    No oracle encode at this size (it would take minutes): run-to-run determinism, every rANS stream decodes with the oracle's
    inverse coder to V·N symbols, and the section is consumed to the last byte; DMI_100M_DEVICE_CHAINS=1 additionally codes the same
    mesh with the device chains forced (≈ 3 s of chain) and compares the two forms byte for byte.  ("rANS chunked across 8 GPUs" is a
    different bitstream — SURVEY F8 — and is not built.)"""
    if _host_memory_gib() < 64:
        pytest.skip("needs ≈ 45 GB of host memory")
    n = int(os.environ.get("DMI_100M_N", "7071"))
    faces, pos, nrm, uv = synth.torus_grid(n)
    atts = [dmi.Attribute(pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION, unique_id=0), dmi.Attribute(nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, unique_id=1, parent_index=0),
            dmi.Attribute(uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, unique_id=2, parent_index=0)]
    mesh = dmi.Mesh(faces, atts)
    job = dmi.mesh_prepare(mesh, dmi.Config(pos_bits=14, flags=dmi.FLAG_TIMINGS))
    a = job.encode()
    b = job.encode()
    assert a == b
    assert job.timings()["host_chains"] == 1
    job.close()
    check_attribute_section(a, n * n)
    if os.environ.get("DMI_100M_DEVICE_CHAINS") == "1":   # ≈ 3 s of device chain: opt-in
        os.environ["DMI_CHAINS"] = "device"
        try:
            j2 = dmi.mesh_prepare(mesh, dmi.Config(pos_bits=14))
        finally:
            del os.environ["DMI_CHAINS"]
        assert j2.encode() == a
        j2.close()


@pytest.mark.parametrize("pos_bits,uv_bits", [(14, 12), (12, 10)])
def test_one_million_triangles_at_14_and_12_bits_byte_parity(pos_bits, uv_bits):
    """configs[4]'s quantization (14-bit positions) at a size the oracle still answers in seconds: 999 698 triangles, pos+nrm+uv, whole
    `.drc` byte for byte (the 100M-triangle mesh itself is property-checked only).  The operand bounds of the sweep's exact tiers (f64
    texture-coordinate projection, 24-bit multiplier for the fan sums) depend on the bit widths; at these the sweep defers nothing."""
    mesh = synth.torus_mesh(707)
    cfg = dmi.Config(pos_bits=pos_bits, uv_bits=uv_bits, flags=dmi.FLAG_TIMINGS)
    want = oracle_from_product_mesh(mesh).encode(pos_bits=pos_bits, uv_bits=uv_bits)
    job = dmi.mesh_prepare(mesh, cfg)
    try:
        got = job.header_and_connectivity + job.encode()
        fixups = job.timings()["texcoord_fixups"]
    finally:
        job.close()
    _assert_same(got, want, f"1M triangles at {pos_bits}/{uv_bits} bits")
    assert fixups == 0
    _assert_same(dmi.encode_mesh(mesh, dmi.Config(pos_bits=pos_bits, uv_bits=uv_bits)), want, f"1M triangles at {pos_bits}/{uv_bits} bits, dmi_encode_mesh")


@pytest.mark.parametrize("n,pos_bits,uv_bits", [(40, 20, 16), (96, 21, 16), (300, 20, 16)])
def test_wide_quantization_provably_enters_the_texcoord_fixup_kernel(n, pos_bits, uv_bits):
    """With 20/21-bit positions on a coarse grid the edge vectors leave the fused sweep's exact f64 tier (|pn| components ≥ 2^15): those
    entries go to k_texcoord_fixup's general i64 form.  The job reports how many (dmi_timings.texcoord_fixups) — the test fails if the
    path it claims to cover was not taken — and the bytes must still be the oracle's."""
    mesh = synth.torus_mesh(n)
    cfg = dmi.Config(pos_bits=pos_bits, uv_bits=uv_bits, flags=dmi.FLAG_TIMINGS)
    want = oracle_from_product_mesh(mesh).encode(pos_bits=pos_bits, uv_bits=uv_bits)
    job = dmi.mesh_prepare(mesh, cfg)
    try:
        got = job.header_and_connectivity + job.encode()
        fixups = job.timings()["texcoord_fixups"]
    finally:
        job.close()
    _assert_same(got, want, f"grid {n} at {pos_bits}/{uv_bits} bits")
    assert fixups > 0, "no entry was deferred: this case no longer reaches k_texcoord_fixup"
