"""Deterministic synthetic meshes for tests and bench.py: a closed torus-topology
grid of n×n quads split into 2 triangles (F = 2n², V = n², valence 6, no boundary, no seams).
Plain numpy workload plumbing — no encoder arithmetic."""
import numpy as np

from .binding import ATT_NORMAL, ATT_POSITION, ATT_TEXCOORD, DOMAIN_CORNER, DOMAIN_POSITION, Attribute, Mesh

SEED = 0xD7AC0


def grid_size_for_triangles(f):
    return int(np.sqrt(f / 2.0))


def torus_grid(n, seed=SEED, normals=True, uvs=True, open_boundary=False):
    """Returns (faces [F,3] uint32, pos [V,3] f32, nrm [V,3] f32 or None, uv [V,2] f32 or None)."""
    rng = np.random.default_rng(seed)
    iu, iv = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    u = (iu.ravel() / n).astype(np.float64)
    v = (iv.ravel() / n).astype(np.float64)
    R, r = 1.0, 0.35
    cu, su, cv, sv = np.cos(2 * np.pi * u), np.sin(2 * np.pi * u), np.cos(2 * np.pi * v), np.sin(2 * np.pi * v)
    pos = np.stack([(R + r * cv) * cu, (R + r * cv) * su, r * sv], axis=1)
    pos += rng.uniform(-1e-3, 1e-3, size=pos.shape)
    pos = pos.astype(np.float32)
    nrm = None
    if normals:
        # analytic torus normal + a little noise (so that no two vertices share a normal), unit length
        nr = np.stack([cv * cu, cv * su, sv], axis=1) + rng.uniform(-1e-3, 1e-3, size=(n * n, 3))
        nrm = (nr / np.linalg.norm(nr, axis=1, keepdims=True)).astype(np.float32)
    uv = None
    if uvs:
        # periodic chart (no jump where the torus closes) + noise.  A plain (u, v) chart puts a 0↔1
        # discontinuity along the closing edges; at 10M triangles its long tail of rare residual symbols
        # drives the reference's frequency normalisation (encode/entropy/rans.rs:172-190) to assign a
        # zero frequency to occurring symbols, after which RansCoder::write never terminates — such a
        # mesh has no reference output to be bit-exact against.
        uv = np.stack([0.5 + 0.49 * np.cos(2 * np.pi * u), 0.5 + 0.49 * np.cos(2 * np.pi * v)], axis=1)
        uv = uv + rng.uniform(-1e-4, 1e-4, size=(n * n, 2))
        uv = np.clip(uv, 0.0, 1.0).astype(np.float32)
    m = n - 1 if open_boundary else n
    a, b = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
    a, b = a.ravel(), b.ravel()
    a1, b1 = (a + 1) % n, (b + 1) % n
    i00, i10, i01, i11 = a * n + b, a1 * n + b, a * n + b1, a1 * n + b1
    faces = np.empty((2 * m * m, 3), np.uint32)
    faces[0::2] = np.stack([i00, i10, i11], axis=1)
    faces[1::2] = np.stack([i00, i11, i01], axis=1)
    return faces, pos, nrm, uv


def torus_mesh(n, seed=SEED, normals=True, uvs=True, open_boundary=False):
    """A `Mesh` whose attributes are already in MeshBuilder's output form.  Values are checked to be
    pairwise distinct so that Attribute::from's dedup (core/attribute/mod.rs:394-452) is the identity."""
    faces, pos, nrm, uv = torus_grid(n, seed, normals, uvs, open_boundary)
    for name, arr in (("pos", pos), ("nrm", nrm), ("uv", uv)):
        if arr is None:
            continue
        a = arr + np.float32(0.0)
        rows = np.ascontiguousarray(a).view(np.dtype((np.void, a.dtype.itemsize * a.shape[1]))).ravel()
        if len(np.unique(rows)) != len(rows):
            # a chance duplicate: go through the builder so that value dedup (and the seams it creates) is applied
            from .binding import MeshBuilder
            b = MeshBuilder()
            pid = b.add_attribute(pos, ATT_POSITION, DOMAIN_POSITION)
            if nrm is not None:
                b.add_attribute(nrm, ATT_NORMAL, DOMAIN_CORNER, parents=[pid])
            if uv is not None:
                b.add_attribute(uv, ATT_TEXCOORD, DOMAIN_CORNER, parents=[pid])
            b.set_connectivity_attribute(faces)
            return b.build()
    atts = [Attribute(pos, ATT_POSITION, DOMAIN_POSITION, unique_id=0)]
    if nrm is not None:
        atts.append(Attribute(nrm, ATT_NORMAL, DOMAIN_CORNER, unique_id=len(atts), parent_index=0))
    if uv is not None:
        atts.append(Attribute(uv, ATT_TEXCOORD, DOMAIN_CORNER, unique_id=len(atts), parent_index=0))
    return Mesh(faces, atts)


def batch_meshes(n_meshes, lo=2e3, hi=2e5, seed=SEED):
    """The batch workload (BASELINE configs[3] shape): n independent torus grids, triangle counts log-uniform in [lo, hi]."""
    rng = np.random.default_rng(seed)
    tris = np.exp(rng.uniform(np.log(lo), np.log(hi), size=n_meshes))
    return [torus_mesh(max(8, grid_size_for_triangles(t)), seed=seed + 7 * k) for k, t in enumerate(tris)]


def torus_glb(n, seed=SEED, open_boundary=False, seams=False):
    """One torus grid as a minimal GLB (one mesh, one triangle primitive; POSITION / NORMAL / TEXCOORD_0 as separate float accessors,
    UNSIGNED_SHORT indices when the vertex count allows, else UNSIGNED_INT) → (glb bytes, triangles).  seams: the exporter-style sheet of
    seam_torus_rows (positions / normals repeated along the closing curves, a UV seam there)."""
    import json
    import struct
    faces, pos, nrm, uv = seam_torus_rows(n, seed) if seams else torus_grid(n, seed, open_boundary=open_boundary)
    idx = faces.ravel().astype("<u2" if len(pos) <= 65535 else "<u4")
    parts = [pos.astype("<f4").tobytes(), nrm.astype("<f4").tobytes(), uv.astype("<f4").tobytes(), idx.tobytes()]
    views, off = [], 0
    for b in parts:
        views.append({"buffer": 0, "byteOffset": off, "byteLength": len(b)})
        off += (len(b) + 3) & ~3
    binary = b"".join(b + b"\0" * ((4 - len(b) % 4) % 4) for b in parts)
    acc = [{"bufferView": 0, "componentType": 5126, "count": len(pos), "type": "VEC3", "min": pos.min(axis=0).tolist(), "max": pos.max(axis=0).tolist()},
           {"bufferView": 1, "componentType": 5126, "count": len(pos), "type": "VEC3"},
           {"bufferView": 2, "componentType": 5126, "count": len(pos), "type": "VEC2"},
           {"bufferView": 3, "componentType": 5123 if idx.dtype.itemsize == 2 else 5125, "count": int(idx.size), "type": "SCALAR"}]
    doc = {"asset": {"version": "2.0"}, "buffers": [{"byteLength": len(binary)}], "bufferViews": views, "accessors": acc,
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "mode": 4}]}],
           "nodes": [{"mesh": 0}], "scenes": [{"nodes": [0]}], "scene": 0}
    js = json.dumps(doc, separators=(",", ":")).encode("utf-8")
    js += b" " * ((4 - len(js) % 4) % 4)
    total = 12 + 8 + len(js) + 8 + len(binary)
    return struct.pack("<4sII", b"glTF", 2, total) + struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(binary), 0x004E4942) + binary, len(faces)


def batch_glbs(n_files, lo=2e3, hi=2e5, seed=SEED, seams=False):
    """The batch workload of batch_meshes as GLB files in memory (BASELINE configs[3]: "1024 glTF/glb meshes through … transcode") →
    ([glb bytes], total triangles).  Same grids, same seeds."""
    rng = np.random.default_rng(seed)
    tris = np.exp(rng.uniform(np.log(lo), np.log(hi), size=n_files))
    out, total = [], 0
    for k, t in enumerate(tris):
        g, f = torus_glb(max(8, grid_size_for_triangles(t)), seed=seed + 7 * k, seams=seams)
        out.append(g)
        total += f
    return out, total


def seam_torus_rows(n, seed=SEED):
    """The torus grid as an exporter writes it: an (n+1)×(n+1) sheet of points whose last row / column REPEAT the positions and normals of the
    first (the surface closes) but carry their own texture coordinates (u or v = 1 instead of 0) — per-point rows + faces.  MeshBuilder merges
    the repeated positions / normals (point → value maps), the UV attribute keeps a seam along both closing curves (an attribute corner
    table of its own: attribute_corner_table.rs:16-137).  → (faces [2n², 3] uint32, pos, nrm, uv per point)."""
    _, pos0, nrm0, _ = torus_grid(n, seed)
    rng = np.random.default_rng(seed + 1)
    m = n + 1
    iu, iv = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
    src = ((iu % n) * n + (iv % n)).ravel()
    pos, nrm = pos0[src], nrm0[src]
    uv = np.stack([iu.ravel() / n, iv.ravel() / n], axis=1) + rng.uniform(-1e-4, 1e-4, size=(m * m, 2)) * 0.0
    uv = uv.astype(np.float32)
    a, b = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    a, b = a.ravel(), b.ravel()
    i00, i10, i01, i11 = a * m + b, (a + 1) * m + b, a * m + b + 1, (a + 1) * m + b + 1
    faces = np.empty((2 * n * n, 3), np.uint32)
    faces[0::2] = np.stack([i00, i10, i11], axis=1)
    faces[1::2] = np.stack([i00, i11, i01], axis=1)
    return faces, pos, nrm, uv
