"""BASELINE configs[4] size on ONE GPU: a single ≈100M-triangle mesh (n = 7071 → 99 998 082 triangles), pos+nrm+uv, 14-bit positions.
No oracle at this size: checks determinism (two encodes), that every rANS stream of the attribute section decodes with the oracle's inverse
coder to exactly V·N symbols, and that the section is consumed to the last byte.  Prints one JSON line.  Needs ≈ 40 GB of host memory."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import draco_oxide_amd as dmi
from draco_oxide_amd import synth
import orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 7071
pos_bits = int(sys.argv[2]) if len(sys.argv) > 2 else 14
t0 = time.time()
faces, pos, nrm, uv = synth.torus_grid(n)
atts = [dmi.Attribute(pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION, unique_id=0), dmi.Attribute(nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, unique_id=1, parent_index=0),
        dmi.Attribute(uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, unique_id=2, parent_index=0)]
mesh = dmi.Mesh(faces, atts)
t_gen = time.time() - t0
import torch  # noqa: E402  (one HIP runtime for torch and the library: torch first)
t0 = time.time()
job = dmi.mesh_prepare(mesh, dmi.Config(pos_bits=pos_bits, flags=dmi.FLAG_TIMINGS))
t_prep = time.time() - t0
t0 = time.time(); a = job.encode(); t_enc = time.time() - t0
tm = job.timings()
t0 = time.time(); b = job.encode(); t_enc2 = time.time() - t0
assert a == b, "two encodes differ"
head = job.header_and_connectivity
job.close()
# whole .drc in one call (device corner tables, host walks, relabelling, encode): twice — the second call has the library's pools warm
e2e = []
for _ in range(2):
    t0 = time.time(); drc = dmi.encode_mesh(mesh, dmi.Config(pos_bits=pos_bits, flags=dmi.FLAG_TIMINGS)); e2e.append(time.time() - t0)
    assert drc == head + a, "dmi_encode_mesh differs from prepare + encode"
call = dmi.last_call_timings()


def leb(buf, p):
    v = s = 0
    while True:
        x = buf[p]; p += 1
        v |= (x & 0x7F) << s; s += 7
        if not x & 0x80:
            return v, p


nA = a[0]
assert nA == 3
p = 1 + 3 * nA + 7 * nA
counts = [n * n * 3, n * n * 2, n * n * 2]
for i in range(nA):
    scheme, transform, rans = a[p], a[p + 1], a[p + 2]
    assert rans == 1 and (scheme, transform) == [(1, 1), (6, 3), (5, 1)][i]
    p += 3
    syms, used = orc.decode_symbols(a[p:p + 1024 * 1024 * 1024], counts[i])
    assert len(syms) == counts[i]
    p += used
    if scheme == 6:
        p += 9; ln, p = leb(a, p); p += ln + 1
    elif scheme == 5:
        p += 5; ln, p = leb(a, p); p += ln + 8 + 13
    else:
        p += 8 + 17
assert p == len(a)
F = len(faces)
print(json.dumps({"triangles": F, "pos_bits": pos_bits, "attribute_section_bytes": len(a), "bytes_per_triangle": round((len(a) + len(job.header_and_connectivity)) / F, 3),
                  "generate_s": round(t_gen, 1), "host_prepare_s": round(t_prep, 1), "encode_s": round(min(t_enc, t_enc2), 3), "mtri_per_s": round(F / min(t_enc, t_enc2) / 1e6, 2),
                  "stages_ms": {k: round(float(tm[k]), 2) for k in ("quantize_ms", "predict_ms", "histogram_ms", "table_ms", "rans_ms", "total_ms")},
                  "end_to_end_s": [round(x, 3) for x in e2e], "end_to_end_mtri_per_s": round(F / min(e2e) / 1e6, 2),
                  "end_to_end_split_ms": {k: round(float(call[k]), 1) for k in ("tables_ms", "connectivity_ms", "job_create_ms", "total_ms", "call_ms")},
                  "roofline_frac_of_8TBps": round(tm["predict_bytes"] / ((tm["quantize_ms"] + tm["predict_ms"]) * 1e-3) / 8e12, 4),
                  "checks": "deterministic; 3 rANS streams decode to V*N symbols; section consumed to the last byte; dmi_encode_mesh == prepare + encode"}))
