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
All primitives of a file — of a whole LIST of files (`transcode_files`, BASELINE configs[3]) — are encoded as ONE batch: one
dmi_jobs_encode on one GPU, one per device from a single process (dmi_jobs_encode_devices), or dealt over the ranks of a
torch.distributed job and gathered on rank 0 (distributed.encode_meshes_sharded).  Inputs: `.glb`, or `.gltf` with external /
data-URI buffers; `_FEATURE_ID_n` attributes (EXT_mesh_features) become Custom u32 corner attributes (decode.rs:2490-2516).
JSON byte-equality with the reference is not part of the bit-exact contract; the embedded .drc blobs are.
"""
import base64
import json
import os
import urllib.parse
import struct

import numpy as np

from .binding import (ATT_CUSTOM, ATT_NORMAL, ATT_POSITION, ATT_TEXCOORD, DOMAIN_CORNER, DOMAIN_POSITION, Config, MeshBuilder, device_count, jobs_encode,
                      jobs_encode_devices, meshes_prepare, meshes_prepare_devices, shard_meshes)

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


def _buffer_of(binary, view):
    """The bytes of the buffer a view points into: `binary` is the GLB BIN chunk (bytes) or the list of a document's buffers."""
    if isinstance(binary, (bytes, bytearray, memoryview)):
        return binary
    return binary[view.get("buffer", 0)]


def _accessor_f32(doc, binary, index):
    """Raw little-endian f32 rows with the bufferView's stride (decode.rs:2277-2309; no normalized-int handling)."""
    acc = doc["accessors"][index]
    view = doc["bufferViews"][acc["bufferView"]]
    n = _COMPONENTS[acc["type"]]
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = view.get("byteStride", 0) or 4 * n
    count = acc["count"]
    raw = np.frombuffer(_buffer_of(binary, view), dtype=np.uint8, count=stride * (count - 1) + 4 * n, offset=start)
    rows = np.lib.stride_tricks.as_strided(raw, shape=(count, 4 * n), strides=(stride, 1))
    return np.ascontiguousarray(rows).view("<f4").reshape(count, n).astype(np.float32)


def _accessor_indices(doc, binary, index):
    acc = doc["accessors"][index]
    view = doc["bufferViews"][acc["bufferView"]]
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    return np.frombuffer(_buffer_of(binary, view), dtype=_INDEX_DTYPE[acc["componentType"]], count=acc["count"], offset=start).astype(np.uint32)


_SCALAR_DTYPE = {5121: np.dtype("u1"), 5123: np.dtype("<u2"), 5125: np.dtype("<u4"), 5126: np.dtype("<f4")}


def _accessor_u32_scalars(doc, binary, index):
    """A `_FEATURE_ID_n` accessor as u32 (decode.rs:2528-2584: UNSIGNED_BYTE / UNSIGNED_SHORT / UNSIGNED_INT widened, FLOAT cast
    `as u32` = truncate, saturate, NaN → 0; the stride defaults to the component size)."""
    acc = doc["accessors"][index]
    view = doc["bufferViews"][acc["bufferView"]]
    dt = _SCALAR_DTYPE.get(acc["componentType"])
    if dt is None:
        raise ValueError(f"unsupported component type {acc['componentType']} for a feature-id attribute")
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = view.get("byteStride", 0) or dt.itemsize
    count = acc["count"]
    raw = np.frombuffer(_buffer_of(binary, view), dtype=np.uint8, count=stride * (count - 1) + dt.itemsize if count else 0, offset=start)
    if count == 0:
        return np.zeros(0, np.uint32)
    rows = np.lib.stride_tricks.as_strided(raw, shape=(count, dt.itemsize), strides=(stride, 1))
    vals = np.ascontiguousarray(rows).view(dt).reshape(count)
    if dt.kind == "f":
        v = np.nan_to_num(vals.astype(np.float64), nan=0.0, posinf=4294967295.0, neginf=0.0)
        return np.clip(np.trunc(v), 0.0, 4294967295.0).astype(np.uint32)
    return vals.astype(np.uint32)


_STANDARD_PREFIXES = ("POSITION", "NORMAL", "TANGENT", "TEXCOORD_", "COLOR_", "JOINTS_", "WEIGHTS_")


def primitive_to_mesh(doc, binary, prim):
    """One triangle primitive → `Mesh` exactly as the reference builds it (decode.rs:2328-2525).  Returns (mesh, names): `names` =
    the glTF attribute names in AttributeId order (POSITION / NORMAL / TEXCOORD_0 sorted by name, then the `_FEATURE_ID_n`)."""
    if prim.get("mode", 4) != 4:
        raise ValueError("only triangle primitives are transcoded")
    # the reference sorts EVERY standard semantic it knows by name (:2410) and hands the position's index in THAT list to the normals /
    # texture coordinates as their parent id (:2414-2425) — but only POSITION, NORMAL and TEXCOORD_0 are added (:2431-2473), with ids in
    # add order.  With a COLOR_n or JOINTS_n in front of POSITION the parent id names the wrong attribute and its encoder panics.
    standard = sorted(k for k in prim["attributes"] if k.startswith(_STANDARD_PREFIXES))
    names = [k for k in standard if k in _SEMANTIC_TYPE]
    if "POSITION" not in names:
        raise ValueError("primitive without POSITION")
    pos_id = standard.index("POSITION")
    if pos_id != names.index("POSITION") and len(names) > 1:
        raise ValueError("the reference hands NORMAL / TEXCOORD_0 a parent id that is not the position attribute for this set of semantics "
                         f"({standard}); its encoder panics on such a primitive")
    b = MeshBuilder()
    count = None
    for name in names:
        rows = _accessor_f32(doc, binary, prim["attributes"][name])
        count = len(rows)
        if name == "POSITION":
            b.add_attribute(rows, ATT_POSITION, DOMAIN_POSITION)
        else:
            b.add_attribute(rows, _SEMANTIC_TYPE[name], DOMAIN_CORNER, parents=[pos_id])
    # EXT_mesh_features ids → Custom u32 corner attributes without parents (:2490-2516).  The reference walks a HashMap here (its order,
    # and so the ids of several feature-id sets, change from run to run); this driver takes them in name order.
    for name in sorted(k for k in prim["attributes"] if k.startswith("_FEATURE_ID_")):
        b.add_attribute(_accessor_u32_scalars(doc, binary, prim["attributes"][name]).reshape(-1, 1), ATT_CUSTOM, DOMAIN_CORNER)
        names.append(name)
    idx = _accessor_indices(doc, binary, prim["indices"]) if "indices" in prim else np.arange(count, dtype=np.uint32)
    b.set_connectivity_attribute(idx[: len(idx) // 3 * 3].reshape(-1, 3))
    return b.build(), names


def load_document(source):
    """`.glb` bytes / path, or a `.gltf` path (external and data-URI buffers resolved) → (doc, buffers): buffers = list of bytes."""
    if isinstance(source, (bytes, bytearray, memoryview)):
        doc, binary = read_glb(bytes(source))
        return doc, [binary]
    data = open(source, "rb").read()
    if data[:4] == b"glTF":
        doc, binary = read_glb(data)
        return doc, [binary]
    doc = json.loads(data.decode("utf-8"))
    base = os.path.dirname(os.path.abspath(source))
    buffers = []
    for buf in doc.get("buffers", []):
        uri = buf.get("uri")
        if uri is None:
            raise ValueError(".gltf buffer without a uri")
        if uri.startswith("data:"):
            buffers.append(base64.b64decode(uri.split(",", 1)[1]))
        else:
            # an external buffer of an untrusted .gltf: percent-decoded, relative, and inside the asset's own directory
            rel = urllib.parse.unquote(uri)
            path = os.path.realpath(os.path.join(base, rel))
            root = os.path.realpath(base)
            if os.path.isabs(rel) or not (path == root or path.startswith(root + os.sep)):
                raise ValueError(f".gltf buffer uri leaves the asset directory: {uri!r}")
            buffers.append(open(path, "rb").read())
    return doc, buffers


def _collect(doc, buffers):
    """The primitives of a document that get compressed: [(prim, names, mesh)] (triangle primitives with POSITION and faces)."""
    prims = []
    for mesh in doc.get("meshes", []):
        for prim in mesh.get("primitives", []):
            if prim.get("mode", 4) != 4 or "POSITION" not in prim.get("attributes", {}):
                continue
            if "KHR_draco_mesh_compression" in prim.get("extensions", {}):
                raise ValueError("KHR_draco_mesh_compression input is not supported (decode.rs:2478-2483)")
            m, names = primitive_to_mesh(doc, buffers, prim)
            if len(m.faces) == 0:
                continue                                                           # encode.rs:934-936
            prims.append((prim, names, m))
    return prims


def _assemble(doc, buffers, prims, blobs):
    """The output GLB of one document: compressed primitives get placeholder accessors + the extension, every other bufferView
    (of any input buffer) is carried over into the single BIN chunk."""
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
            chunk = _buffer_of(buffers, v)[start: start + v["byteLength"]]
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
        by_id = sorted(m.attributes, key=lambda a: a.unique_id)                   # AttributeId = add order = `names` order
        ext = {"bufferView": len(new_views) - 1, "attributes": {n: int(a.unique_id) for n, a in zip(names, by_id)}}
        prim.setdefault("extensions", {})["KHR_draco_mesh_compression"] = ext
        if "indices" in prim:
            ia = doc["accessors"][prim["indices"]]
            ia["count"] = int(len(m.faces) * 3)
        for n in names:
            doc["accessors"][prim["attributes"][n]]["count"] = int(m.attributes[0].num_points)
    doc["bufferViews"] = new_views
    doc["buffers"] = [{"byteLength": len(new_bin)}]
    if prims:
        for key in ("extensionsUsed", "extensionsRequired"):
            lst = doc.setdefault(key, [])
            if "KHR_draco_mesh_compression" not in lst:
                lst.append("KHR_draco_mesh_compression")
    return write_glb(doc, bytes(new_bin))


def encode_batch(meshes, cfg=None, devices=None, group=None, device=None):
    """Every mesh of a transcode job as ONE batch → list of `.drc` blobs in mesh order (None on the ranks that are not the
    destination of a sharded job).  torch.distributed initialised with more than one rank: the batch is dealt over the ranks by
    triangle count and gathered on rank 0 (RCCL for an nccl group).  Otherwise `devices` (a count, or "all") spreads it over the
    GPUs of this process (dmi_shard_meshes + dmi_meshes_prepare_devices + dmi_jobs_encode_devices); default: one GPU."""
    if not meshes:
        return []
    try:
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    except ImportError:
        world = 1
    if world > 1:
        from . import distributed
        return distributed.encode_meshes_sharded(meshes, cfg, device=device, group=group)
    n_dev = device_count() if devices == "all" else int(devices or 1)
    n_dev = max(1, min(n_dev, device_count()))
    jobs = []
    try:   # (the jobs hold device memory: closed whatever the encode does)
        if n_dev > 1:
            deal = shard_meshes(meshes, n_dev)
            jobs = meshes_prepare_devices(meshes, deal, cfg)
            sections = jobs_encode_devices(jobs)
        else:
            jobs = meshes_prepare(meshes, cfg or Config.default())   # connectivity stage of all primitives: host walks on a thread pool, tables on the device
            sections = jobs_encode(jobs)
        return [j.header_and_connectivity + s for j, s in zip(jobs, sections)]
    finally:
        for j in jobs:
            j.close()


def transcode_files(sources, cfg=None, devices=None, group=None, device=None):
    """BASELINE configs[3]: a LIST of glTF assets (GLB bytes, `.glb` / `.gltf` paths) → their Draco-compressed GLBs.  The triangle
    primitives of ALL files form one batch (encode_batch: one GPU, several GPUs of this process, or the ranks of a torch.distributed
    job); every file is then reassembled around its blobs (io/gltf/transcoder.rs:134-151 runs the files one by one, and
    io/gltf/encode.rs:1827-1842 their primitives one by one).  Returns [(glb_bytes, [blob, ...]), ...] in input order — on the
    destination rank; None on the other ranks of a sharded job."""
    docs = [load_document(src) for src in sources]
    per_file = [_collect(doc, buffers) for doc, buffers in docs]
    meshes = [m for prims in per_file for (_, _, m) in prims]
    blobs = encode_batch(meshes, cfg, devices=devices, group=group, device=device)
    if blobs is None:
        return None
    out, at = [], 0
    for (doc, buffers), prims in zip(docs, per_file):
        mine = blobs[at: at + len(prims)]
        at += len(prims)
        out.append((_assemble(doc, buffers, prims, mine), [bytes(b) for b in mine]))
    return out


def transcode_glb(data, cfg=None):
    """GLB bytes in → (GLB bytes, blobs) out, every triangle primitive Draco-compressed on the GPU as one batch."""
    (out, blobs), = transcode_files([data], cfg)
    return out, blobs


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
