"""GLB → GLB transcode with KHR_draco_mesh_compression — container plumbing only.

Mirrors what the reference's transcoder does around the hot path (paths relative to draco-oxide/src/):
  * io/gltf/decode.rs:2328-2525  one Mesh per triangle primitive; attributes taken in sorted semantic order
    (NORMAL, POSITION, TEXCOORD_0 → ids 0, 1, 2), accessors read as raw little-endian f32 with the view's
    stride, NORMAL/TEXCOORD get AttributeDomain::Corner with parents = [position id]; MeshBuilder then
    swaps Position to slot 0 (core/mesh/builder.rs:115-125)
  * io/gltf/encode.rs:932-1097   one `encode::encode(mesh, Config::default())` per primitive, blob appended
    to the BIN chunk and zero-padded to 4 bytes (the bufferView byteLength includes the pad), placeholder
    accessors without bufferView, extension attributes POSITION→1, NORMAL→0, TEXCOORD_0→2
  * io/gltf/encode.rs:362-400    GLB container: "glTF", 2, length | JSON chunk (space padded) | BIN chunk
All primitives of a file are encoded as ONE batch (dmi_jobs_encode).  JSON byte-equality with the reference
is not part of the bit-exact contract; the embedded .drc blobs are.
"""
import json
import struct

import numpy as np

from .binding import (ATT_NORMAL, ATT_POSITION, ATT_TEXCOORD, DOMAIN_CORNER, DOMAIN_POSITION, Config, MeshBuilder, jobs_encode, meshes_prepare)

_COMPONENTS = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}
_INDEX_DTYPE = {5121: np.uint8, 5123: np.uint16, 5125: np.uint32}
_SEMANTIC_TYPE = {"POSITION": ATT_POSITION, "NORMAL": ATT_NORMAL, "TEXCOORD_0": ATT_TEXCOORD}


def read_glb(data):
    magic, version, length = struct.unpack_from("<4sII", data, 0)
    if magic != b"glTF" or version != 2:
        raise ValueError("not a GLB v2 file")
    off, doc, binary = 12, None, b""
    while off < length:
        clen, ctype = struct.unpack_from("<II", data, off)
        chunk = data[off + 8: off + 8 + clen]
        if ctype == 0x4E4F534A:
            doc = json.loads(chunk.decode("utf-8"))
        elif ctype == 0x004E4942:
            binary = bytes(chunk)
        off += 8 + clen
    return doc, binary


def write_glb(doc, binary):
    js = json.dumps(doc, separators=(",", ":")).encode("utf-8")
    js += b" " * ((4 - len(js) % 4) % 4)                      # JSON chunk is space padded (encode.rs:392-396)
    binary = bytes(binary) + b"\0" * ((4 - len(binary) % 4) % 4)
    total = 12 + 8 + len(js) + (8 + len(binary) if binary else 0)
    out = struct.pack("<4sII", b"glTF", 2, total) + struct.pack("<II", len(js), 0x4E4F534A) + js
    if binary:
        out += struct.pack("<II", len(binary), 0x004E4942) + binary
    return out


def _accessor_f32(doc, binary, index):
    """Raw little-endian f32 rows with the bufferView's stride (decode.rs:2277-2309; no normalized-int handling)."""
    acc = doc["accessors"][index]
    view = doc["bufferViews"][acc["bufferView"]]
    n = _COMPONENTS[acc["type"]]
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = view.get("byteStride", 0) or 4 * n
    count = acc["count"]
    raw = np.frombuffer(binary, dtype=np.uint8, count=stride * (count - 1) + 4 * n, offset=start)
    rows = np.lib.stride_tricks.as_strided(raw, shape=(count, 4 * n), strides=(stride, 1))
    return np.ascontiguousarray(rows).view("<f4").reshape(count, n).astype(np.float32)


def _accessor_indices(doc, binary, index):
    acc = doc["accessors"][index]
    view = doc["bufferViews"][acc["bufferView"]]
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    return np.frombuffer(binary, dtype=_INDEX_DTYPE[acc["componentType"]], count=acc["count"], offset=start).astype(np.uint32)


def primitive_to_mesh(doc, binary, prim):
    """One triangle primitive → `Mesh` exactly as the reference builds it (decode.rs:2328-2525)."""
    if prim.get("mode", 4) != 4:
        raise ValueError("only triangle primitives are transcoded")
    names = sorted(k for k in prim["attributes"] if k in _SEMANTIC_TYPE)        # sorted by semantic name (:2410)
    if "POSITION" not in names:
        raise ValueError("primitive without POSITION")
    pos_id = names.index("POSITION")
    b = MeshBuilder()
    count = None
    for name in names:
        rows = _accessor_f32(doc, binary, prim["attributes"][name])
        count = len(rows)
        if name == "POSITION":
            b.add_attribute(rows, ATT_POSITION, DOMAIN_POSITION)
        else:
            b.add_attribute(rows, _SEMANTIC_TYPE[name], DOMAIN_CORNER, parents=[pos_id])
    idx = _accessor_indices(doc, binary, prim["indices"]) if "indices" in prim else np.arange(count, dtype=np.uint32)
    b.set_connectivity_attribute(idx[: len(idx) // 3 * 3].reshape(-1, 3))
    return b.build(), names


def transcode_glb(data, cfg=None):
    """GLB bytes in → GLB bytes out, every triangle primitive Draco-compressed on the GPU as one batch."""
    doc, binary = read_glb(data)
    cfg = cfg or Config.default()
    prims, meshes = [], []
    for mesh in doc.get("meshes", []):
        for prim in mesh.get("primitives", []):
            if prim.get("mode", 4) != 4 or "POSITION" not in prim.get("attributes", {}):
                continue
            m, names = primitive_to_mesh(doc, binary, prim)
            if len(m.faces) == 0:
                continue                                                           # encode.rs:934-936
            prims.append((prim, names, m))
            meshes.append(m)
    jobs = meshes_prepare(meshes, cfg)   # host connectivity of all primitives on a thread pool
    sections = jobs_encode(jobs) if jobs else []
    blobs = [j.header_and_connectivity + s for j, s in zip(jobs, sections)]
    for j in jobs:
        j.close()

    # accessors owned by compressed primitives become placeholders; every other bufferView is carried over
    replaced = set()
    for prim, names, _ in prims:
        replaced.update(prim["attributes"][n] for n in names)
        if "indices" in prim:
            replaced.add(prim["indices"])
    new_bin = bytearray()
    new_views, view_map = [], {}

    def carry(view_index):
        if view_index not in view_map:
            v = dict(doc["bufferViews"][view_index])
            start = v.get("byteOffset", 0)
            chunk = binary[start: start + v["byteLength"]]
            v["byteOffset"] = len(new_bin)
            v["buffer"] = 0
            new_bin.extend(chunk)
            new_bin.extend(b"\0" * ((4 - len(new_bin) % 4) % 4))
            view_map[view_index] = len(new_views)
            new_views.append(v)
        return view_map[view_index]

    for i, acc in enumerate(doc.get("accessors", [])):
        if i in replaced:
            acc.pop("bufferView", None)
            acc.pop("byteOffset", None)
        elif "bufferView" in acc:
            acc["bufferView"] = carry(acc["bufferView"])
    for img in doc.get("images", []):
        if "bufferView" in img:
            img["bufferView"] = carry(img["bufferView"])
    for (prim, names, m), blob in zip(prims, blobs):
        start = len(new_bin)
        new_bin.extend(blob)
        new_bin.extend(b"\0" * ((4 - len(new_bin) % 4) % 4))
        new_views.append({"buffer": 0, "byteOffset": start, "byteLength": len(new_bin) - start})    # length includes the pad
        ids = {a.att_type: a.unique_id for a in m.attributes}
        ext = {"bufferView": len(new_views) - 1,
               "attributes": {n: int(ids[_SEMANTIC_TYPE[n]]) for n in names}}
        prim.setdefault("extensions", {})["KHR_draco_mesh_compression"] = ext
        if "indices" in prim:
            ia = doc["accessors"][prim["indices"]]
            ia["count"] = int(len(m.faces) * 3)
        for n in names:
            doc["accessors"][prim["attributes"][n]]["count"] = int(m.attributes[0].num_points)
    doc["bufferViews"] = new_views
    doc["buffers"] = [{"byteLength": len(new_bin)}]
    for key in ("extensionsUsed", "extensionsRequired"):
        lst = doc.setdefault(key, [])
        if "KHR_draco_mesh_compression" not in lst:
            lst.append("KHR_draco_mesh_compression")
    return write_glb(doc, bytes(new_bin)), blobs


def draco_blobs_of(glb):
    """The KHR_draco_mesh_compression payloads of a GLB (for tests)."""
    doc, binary = read_glb(glb)
    out = []
    for mesh in doc.get("meshes", []):
        for prim in mesh.get("primitives", []):
            ext = prim.get("extensions", {}).get("KHR_draco_mesh_compression")
            if ext:
                v = doc["bufferViews"][ext["bufferView"]]
                out.append((binary[v.get("byteOffset", 0): v.get("byteOffset", 0) + v["byteLength"]], ext["attributes"]))
    return out
