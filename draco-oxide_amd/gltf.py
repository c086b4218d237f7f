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
All primitives of a file — of a whole LIST of files (`transcode_files`, BASELINE configs[3]) — go through the device as batches: the
accessors' bytes are handed to dmi_meshes_build as they lie in the BIN chunk (MeshBuilder::build on the GPU for all primitives at once —
nothing is copied or deduplicated in Python), dmi_built_meshes_prepare runs the connectivity stage on the resident result and
dmi_jobs_encode codes every stream; consecutive stages overlap (build of stage k+2, prepare of stage k+1, encode of stage k).  On one
GPU, on several GPUs of this process (a share per device, a thread each), or dealt over the ranks of a torch.distributed job BEFORE
anything is built (a rank parses the JSON of every file but touches only the bytes of its own primitives) and gathered on rank 0.  Inputs: `.glb`, or `.gltf` with external /
data-URI buffers; `_FEATURE_ID_n` attributes (EXT_mesh_features) become Custom u32 corner attributes (decode.rs:2490-2516).
JSON byte-equality with the reference is not part of the bit-exact contract; the embedded .drc blobs are.
"""
import base64
import json
import os
import sys
import urllib.parse
import struct

import numpy as np

import threading

from .binding import (DracoMiError, jobs_encode_raw, transcode_assets, ATT_CUSTOM, ATT_NORMAL, ATT_POSITION, ATT_TEXCOORD, DOMAIN_CORNER, DOMAIN_POSITION, Config, MeshBuilder, RawMesh, built_meshes_prepare, device_count,
                      jobs_encode, jobs_encode_devices, last_build_timings, meshes_build, meshes_prepare, meshes_prepare_devices, shard_meshes, thread_host_threads)

_COMPONENTS = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4}
_INDEX_DTYPE = {5121: np.uint8, 5123: np.uint16, 5125: np.uint32}
_SEMANTIC_TYPE = {"POSITION": ATT_POSITION, "NORMAL": ATT_NORMAL, "TEXCOORD_0": ATT_TEXCOORD}


def read_glb(data, copy=True):
    """GLB container → (JSON document, BIN chunk).  copy=False: the BIN chunk is a memoryview of `data` (a transcode reads the accessors
    where the caller's bytes lie; a GB of BIN chunks is not copied to be parsed)."""
    magic, version, length = struct.unpack_from("<4sII", data, 0)
    if magic != b"glTF" or version != 2:
        raise ValueError("not a GLB v2 file")
    off, doc, binary = 12, None, b""
    view = memoryview(data)
    while off < length:
        clen, ctype = struct.unpack_from("<II", data, off)
        chunk = view[off + 8: off + 8 + clen]
        if ctype == 0x4E4F534A:
            doc = json.loads(bytes(chunk).decode("utf-8"))
        elif ctype == 0x004E4942:
            binary = bytes(chunk) if copy else chunk
        off += 8 + clen
    return doc, binary


def write_glb(doc, binary, with_bin_offset=False):
    """GLB container around `doc` and the BIN chunk — bytes, or a list of byte pieces (any buffers; joined here, ONE copy of the chunk).
    with_bin_offset: returns (glb, offset of the chunk's payload in it)."""
    js = json.dumps(doc, separators=(",", ":")).encode("utf-8")
    js_pad = b" " * ((4 - len(js) % 4) % 4)                   # JSON chunk is space padded (encode.rs:392-396)
    pieces = list(binary) if isinstance(binary, (list, tuple)) else [binary]
    bin_len = sum(len(p) for p in pieces)
    bin_pad = b"\0" * ((4 - bin_len % 4) % 4)
    n_js, n_bin = len(js) + len(js_pad), bin_len + len(bin_pad)
    total = 12 + 8 + n_js + (8 + n_bin if n_bin else 0)
    parts = [struct.pack("<4sII", b"glTF", 2, total), struct.pack("<II", n_js, 0x4E4F534A), js, js_pad]
    if n_bin:
        parts += [struct.pack("<II", n_bin, 0x004E4942)] + pieces + [bin_pad]
    glb = b"".join(parts)
    return (glb, 12 + 8 + n_js + 8) if with_bin_offset else glb


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


def _accessor_f32_view(doc, binary, index):
    """_accessor_f32 without the copy: a (count, n) float32 VIEW of the buffer's bytes (rows `byteStride` apart)."""
    acc = doc["accessors"][index]
    view = doc["bufferViews"][acc["bufferView"]]
    n = _COMPONENTS[acc["type"]]
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = view.get("byteStride", 0) or 4 * n
    count = acc["count"]
    if count == 0:
        return np.zeros((0, n), np.float32)
    if stride == 4 * n:
        return np.frombuffer(_buffer_of(binary, view), dtype="<f4", count=count * n, offset=start).reshape(count, n)
    raw = np.frombuffer(_buffer_of(binary, view), dtype=np.uint8, count=stride * (count - 1) + 4 * n, offset=start)
    rows = np.lib.stride_tricks.as_strided(raw, shape=(count, 4 * n), strides=(stride, 1), writeable=False)
    return rows.view("<f4")


def _accessor_indices_view(doc, binary, index):
    acc = doc["accessors"][index]
    view = doc["bufferViews"][acc["bufferView"]]
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    return np.frombuffer(_buffer_of(binary, view), dtype=_INDEX_DTYPE[acc["componentType"]], count=acc["count"], offset=start)


def _primitive_names(prim):
    """The attribute names of a triangle primitive in AttributeId order, and the position's id — or a ValueError where the reference's
    importer hands out a wrong parent id (see primitive_to_mesh)."""
    standard = sorted(k for k in prim["attributes"] if k.startswith(_STANDARD_PREFIXES))
    names = [k for k in standard if k in _SEMANTIC_TYPE]
    if "POSITION" not in names:
        raise ValueError("primitive without POSITION")
    pos_id = standard.index("POSITION")
    if pos_id != names.index("POSITION") and len(names) > 1:
        raise ValueError("the reference hands NORMAL / TEXCOORD_0 a parent id that is not the position attribute for this set of semantics "
                         f"({standard}); its encoder panics on such a primitive")
    return names, pos_id


def primitive_weight(doc, prim):
    """Triangles of a primitive from the JSON alone (what a sharded job deals by, before any rank touches the BIN chunk)."""
    if "indices" in prim:
        return int(doc["accessors"][prim["indices"]]["count"]) // 3
    return int(doc["accessors"][prim["attributes"]["POSITION"]]["count"]) // 3


def primitive_to_raw(doc, binary, prim):
    """One triangle primitive → (RawMesh, names): what primitive_to_mesh feeds MeshBuilder (decode.rs:2328-2525), as VIEWS of the
    accessors' bytes — dmi_meshes_build runs MeshBuilder::build on the device."""
    if prim.get("mode", 4) != 4:
        raise ValueError("only triangle primitives are transcoded")
    names, pos_id = _primitive_names(prim)
    rm = RawMesh()
    count = 0
    for name in names:
        rows = _accessor_f32_view(doc, binary, prim["attributes"][name])
        count = len(rows)
        if name == "POSITION":
            rm.add_attribute(rows, ATT_POSITION, DOMAIN_POSITION)
        else:
            rm.add_attribute(rows, _SEMANTIC_TYPE[name], DOMAIN_CORNER, parents=[pos_id])
    for name in sorted(k for k in prim["attributes"] if k.startswith("_FEATURE_ID_")):
        rm.add_attribute(_accessor_u32_scalars(doc, binary, prim["attributes"][name]).reshape(-1, 1), ATT_CUSTOM, DOMAIN_CORNER)
        names.append(name)
    rm.set_indices(_accessor_indices_view(doc, binary, prim["indices"]) if "indices" in prim else np.arange(count, dtype=np.uint32))
    return rm, names


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
        doc, binary = read_glb(source, copy=False)
        return doc, [binary]
    data = open(source, "rb").read()
    if data[:4] == b"glTF":
        doc, binary = read_glb(data, copy=False)
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


def _plan(doc):
    """The primitives of a document that get compressed — [(prim, names, triangles)] — from the JSON alone."""
    prims = []
    for mesh in doc.get("meshes", []):
        for prim in mesh.get("primitives", []):
            if prim.get("mode", 4) != 4 or "POSITION" not in prim.get("attributes", {}):
                continue
            if "KHR_draco_mesh_compression" in prim.get("extensions", {}):
                raise ValueError("KHR_draco_mesh_compression input is not supported (decode.rs:2478-2483)")
            names, _ = _primitive_names(prim)
            names = names + sorted(k for k in prim["attributes"] if k.startswith("_FEATURE_ID_"))
            prims.append((prim, names, primitive_weight(doc, prim)))
    return prims


def _collect(doc, buffers):
    """Host-builder form (tests, encode_batch callers): [(prim, names, mesh)] of the primitives that get compressed."""
    out = []
    for prim, _, _ in _plan(doc):
        m, names = primitive_to_mesh(doc, buffers, prim)
        if len(m.faces) == 0:
            continue                                                           # encode.rs:934-936
        out.append((prim, names, m))
    return out


def _accessor_users(doc):
    """How many places of the document name each accessor: primitive attributes, indices and morph targets of EVERY primitive, animation samplers, skins."""
    users = {}

    def use(v):
        if isinstance(v, int) and not isinstance(v, bool) and v >= 0:
            users[v] = users.get(v, 0) + 1
    for mesh in doc.get("meshes", []):
        for prim in mesh.get("primitives", []):
            for v in prim.get("attributes", {}).values():
                use(v)
            use(prim.get("indices"))
            for tgt in prim.get("targets", []) or []:
                if isinstance(tgt, dict):
                    for v in tgt.values():
                        use(v)
    for anim in doc.get("animations", []) or []:
        for smp in anim.get("samplers", []) or []:
            use(smp.get("input"))
            use(smp.get("output"))
    for skin in doc.get("skins", []) or []:
        use(skin.get("inverseBindMatrices"))
    return users


def _privatize_accessors(doc, prims, results):
    """A compressed primitive's accessors become placeholders (no bufferView, new counts).  An accessor that something ELSE names too — a primitive that stays
    uncompressed (another mode, no face left), a second compressed primitive, a morph target, an animation — must keep its data / must not take another
    primitive's counts: the compressed primitive gets a copy of its own at the end of the accessor list (the reference writes fresh accessors per primitive:
    encode.rs:958-1097).  Same walk, same order as csrc/dmi_gltf.cpp privatize_accessors."""
    users = _accessor_users(doc)
    accessors = doc.get("accessors", [])
    for (prim, names, *_), res in zip(prims, results):
        if res is None:
            continue
        refs = [("attributes", n) for n in names] + ([("indices", None)] if "indices" in prim else [])
        for where, n in refs:
            ai = prim["attributes"][n] if where == "attributes" else prim["indices"]
            if not isinstance(ai, int) or isinstance(ai, bool) or not 0 <= ai < len(accessors) or users.get(ai, 0) <= 1:
                continue
            users[ai] -= 1
            accessors.append(json.loads(json.dumps(accessors[ai])))
            if where == "attributes":
                prim["attributes"][n] = len(accessors) - 1
            else:
                prim["indices"] = len(accessors) - 1


def _assemble(doc, buffers, prims, results):
    """The output GLB of one document.  prims = [(prim, names, …)], results = [(blob, num_faces, num_points) or None] per primitive
    (None: the built mesh has no face — encode.rs:934-936 leaves such a primitive alone).  Compressed primitives get placeholder
    accessors + the extension, every other bufferView (of any input buffer) is carried over into the single BIN chunk.  A blob may be a tuple of
    pieces (header + connectivity bytes, a view of the library-owned attribute section): they meet for the first time in the output file.
    Returns (glb bytes, [blob, ...]) — the blobs as views into the GLB (no second copy of them)."""
    _privatize_accessors(doc, prims, results)
    replaced = set()
    for (prim, names, *_), res in zip(prims, results):
        if res is None:
            continue
        replaced.update(prim["attributes"][n] for n in names)
        if "indices" in prim:
            replaced.add(prim["indices"])
    pieces, size = [], [0]                                                      # the new BIN chunk as a list of byte pieces (joined once, by write_glb)
    new_views, view_map = [], {}
    pad = (b"", b"\0", b"\0\0", b"\0\0\0")

    def put(chunk):
        if isinstance(chunk, tuple):
            for c in chunk:
                pieces.append(c)
                size[0] += len(c)
        else:
            pieces.append(chunk)
            size[0] += len(chunk)
        if size[0] % 4:
            p = pad[4 - size[0] % 4]
            pieces.append(p)
            size[0] += len(p)

    def carry(view_index):
        if view_index not in view_map:
            v = dict(doc["bufferViews"][view_index])
            start = v.get("byteOffset", 0)
            chunk = _buffer_of(buffers, v)[start: start + v["byteLength"]]
            v["byteOffset"] = size[0]
            v["buffer"] = 0
            put(chunk)
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
    any_compressed = False
    spans = []                                                                   # (start, end) of every blob in the BIN chunk
    for (prim, names, *_), res in zip(prims, results):
        if res is None:
            continue
        blob, num_faces, num_points = res
        any_compressed = True
        start = size[0]
        put(blob)
        spans.append((start, start + (sum(len(c) for c in blob) if isinstance(blob, tuple) else len(blob))))
        new_views.append({"buffer": 0, "byteOffset": start, "byteLength": size[0] - start})    # length includes the pad
        # AttributeId = add order = `names` order (the built mesh has Position in slot 0, ids unchanged: builder.rs:115-125)
        ext = {"bufferView": len(new_views) - 1, "attributes": {n: k for k, n in enumerate(names)}}
        prim.setdefault("extensions", {})["KHR_draco_mesh_compression"] = ext
        if "indices" in prim:
            doc["accessors"][prim["indices"]]["count"] = int(num_faces) * 3
        for n in names:
            doc["accessors"][prim["attributes"][n]]["count"] = int(num_points)
    doc["bufferViews"] = new_views
    doc["buffers"] = [{"byteLength": size[0]}]
    if any_compressed:
        for key in ("extensionsUsed", "extensionsRequired"):
            lst = doc.setdefault(key, [])
            if "KHR_draco_mesh_compression" not in lst:
                lst.append("KHR_draco_mesh_compression")
    glb, at = write_glb(doc, pieces, with_bin_offset=True)
    mv = memoryview(glb)
    return glb, [mv[at + a: at + b] for a, b in spans]


def _chunks_by_weight(weights, limit, ramp=False):
    """Consecutive index ranges of ≈ `limit` triangles each.  ramp: the first range is a third of that (a pipeline's first stage runs alone:
    the sooner it is through, the sooner the stages overlap)."""
    out, cur, acc = [], [], 0
    ramp = ramp and os.environ.get("DMI_PIPELINE_RAMP", "1") != "0"
    for i, w in enumerate(weights):
        if cur and acc + w > (limit // 3 if ramp and not out else limit):
            out.append(cur)
            cur, acc = [], 0
        cur.append(i)
        acc += w
    if cur:
        out.append(cur)
    return out


def _pipelined(chunks, *stages):
    """stages[-1](…stages[0](chunk)) for every chunk, every stage on a thread of its own, one chunk in flight between neighbours: stage s of
    chunk k+1 runs beside stage s+1 of chunk k (the library calls release the GIL).  A stage given as (fn, workers) runs on that many threads
    (its chunks may then finish out of order: the stages after it must not care).  The first element of what a stage returns is a list of
    jobs or a built batch (released if a later stage fails)."""
    stages = [st if isinstance(st, tuple) else (st, 1) for st in stages]
    if len(chunks) <= 1 or len(stages) == 1:
        for ch in chunks:
            x = ch
            for fn, _ in stages:
                x = fn(x)
        return
    import queue
    qs = [queue.Queue(maxsize=1) for _ in range(len(stages) - 1)]
    err = []
    stop = object()
    lock = threading.Lock()
    alive = [w for _, w in stages]

    def run(si):
        src = qs[si - 1]
        fn = stages[si][0]
        while True:
            x = src.get()
            if x is stop:
                src.put(stop)                                              # (for the stage's other workers; nothing follows a stop)
                break
            try:
                if err:
                    _drop(x)
                    continue
                y = fn(x)
                if si + 1 < len(stages):
                    qs[si].put(y)
            except BaseException as e:                                    # noqa: BLE001 — re-raised on the caller's thread
                err.append(e)
                _drop(x)
        with lock:
            alive[si] -= 1
            last = alive[si] == 0
        if last and si + 1 < len(stages):
            qs[si].put(stop)

    threads = [threading.Thread(target=run, args=(si,)) for si in range(1, len(stages)) for _ in range(stages[si][1])]
    for t in threads:
        t.start()
    try:
        for ch in chunks:
            if err:
                break
            qs[0].put(stages[0][0](ch))
    except BaseException as e:                                            # noqa: BLE001
        err.append(e)
    qs[0].put(stop)
    for t in threads:
        t.join()
    if err:
        raise err[0]


def _drop(mid):
    """Release what a stage handed on when a later stage cannot take it: jobs are closed, a built batch is freed."""
    first = mid[0] if isinstance(mid, tuple) and mid else None
    if hasattr(first, "free"):
        first.free()
    elif isinstance(first, list):
        for j in first:
            if hasattr(j, "close"):
                j.close()


PIPELINE_TRIANGLES = int(os.environ.get("DMI_PIPELINE_TRIANGLES", 0))     # triangles per pipeline stage; 0 = by the size of the batch (stage_triangles)


def stage_triangles(total):
    """Stage size of a pipelined batch: about four stages (enough to overlap build / prepare / encode / reassembly), between 3M and 12M triangles
    (every stage pays fixed costs — the chain launch is bounded by its longest stream, ≈ 5 ms — and a stage above ≈ 16M stops overlapping;
    measured on 1024 GLBs / 45M triangles: 2M 153, 4M 208, 6M 233, 9M 244, 12M 247, 16M 249, 24M 210 Mtri/s)."""
    return PIPELINE_TRIANGLES or min(12 << 20, max(3 << 20, total // 4))


def encode_raw_batch(raws, cfg=None, pipeline=True, timings=None, weights=None, on_done=None, keep=None):
    """RawMesh list → [(blob, num_faces, num_points) or None (no face left)] on ONE device: dmi_meshes_build → dmi_built_meshes_prepare →
    dmi_jobs_encode, in stages of stage_triangles() triangles: the build of stage k+2, the prepare of stage k+1 and the encode of stage k
    run side by side (three threads; the host walks of the prepare are the longest step and keep the host's cores, the build's packing and
    the encode's read-back fit beside them).  `raws` may be a callable i → RawMesh (made when its stage is built: the accessor views of
    stage k+2 are set up beside the device work of the earlier stages) together with `weights` (triangles per primitive).
    on_done(indices, out) (optional): a fourth stage, called with the primitives of every finished stage (the caller reassembles files).
    keep (optional list): zero-copy form — a blob is then the tuple (header + connectivity bytes, uint8 view of the library-owned attribute section) and
    the stage's EncodedBatch is appended to `keep`: the caller frees them once the views have been used (two 80 MB copies under the interpreter lock less)."""
    cfg = cfg or Config.default()
    make = raws if callable(raws) else None
    n = len(weights) if make else len(raws)
    out = [None] * n
    if not n:
        return out
    import time
    if weights is None:
        weights = [len(r.indices) // 3 if r.indices is not None else 0 for r in raws]
    chunks = _chunks_by_weight(weights, stage_triangles(sum(weights)), ramp=True) if pipeline else [list(range(n))]
    tm = timings if timings is not None else {}
    for key in ("views_s", "build_s", "prepare_s", "encode_s", "build_kernels_ms", "build_pack_ms"):
        tm.setdefault(key, 0.0)

    trace = tm.get("trace")                                               # (a list: (step, first primitive, start, end) per stage and step, seconds)

    def build(ch):
        t0 = time.perf_counter()
        mine = [make(i) for i in ch] if make else [raws[i] for i in ch]
        t1 = time.perf_counter()
        batch = meshes_build(mine, cfg)
        bt = last_build_timings()
        tm["views_s"] += t1 - t0
        tm["build_s"] += time.perf_counter() - t1
        tm["build_kernels_ms"] += bt["kernels_ms"]
        tm["build_pack_ms"] += bt["pack_ms"]
        if trace is not None:
            trace.append(("build", ch[0], t0, time.perf_counter()))
        return batch, ch

    def prepare(mid):
        batch, ch = mid
        t0 = time.perf_counter()
        try:
            nf, npts = batch.counts()
            keep = [k for k in range(len(ch)) if nf[k] > 0]
            info = [(int(nf[k]), int(npts[k]), None) for k in keep]
            jobs = built_meshes_prepare(batch, keep, cfg)
        finally:
            batch.free()
        tm["prepare_s"] += time.perf_counter() - t0
        if trace is not None:
            trace.append(("prepare", ch[0], t0, time.perf_counter()))
        return jobs, [ch[k] for k in keep], info

    def encode(mid):
        jobs, where, info = mid[:3]
        t0 = time.perf_counter()
        try:
            if keep is not None and jobs:
                raw = jobs_encode_raw(jobs)
                keep.append(raw)
                for k, (j, i, (nf, npts, _)) in enumerate(zip(jobs, where, info)):
                    out[i] = ((j.header_and_connectivity, raw.view(k)), nf, npts)
            else:
                sections = jobs_encode(jobs) if jobs else []
                for j, s, i, (nf, npts, _) in zip(jobs, sections, where, info):
                    out[i] = (j.header_and_connectivity + s, nf, npts)
        finally:
            for j in jobs:
                j.close()
        tm["encode_s"] += time.perf_counter() - t0
        if trace is not None:
            trace.append(("encode", mid[3][0] if len(mid) > 3 and mid[3] else -1, t0, time.perf_counter()))
        return None, mid[3]

    def prepare_ch(mid):
        return prepare(mid) + (mid[1],)

    def finish(mid):
        t0 = time.perf_counter()
        on_done(mid[1], out)
        if trace is not None:
            trace.append(("assemble", mid[1][0] if mid[1] else -1, t0, time.perf_counter()))
        return None

    _pipelined(chunks, build, prepare_ch, encode, *([finish] if on_done else []))
    return out


_ENCODE_RAW_BATCH = encode_raw_batch   # (a test's stand-in for encode_raw_batch must keep being used by the sharded driver)


def _world_size(group=None):
    """Ranks of the torch.distributed job this process belongs to — 1 when nobody has imported torch.distributed (never imported HERE: the library does not
    need torch, and a process without it runs on the system's HIP runtime instead of the one the torch wheel bundles — binding.load_library)."""
    dist = sys.modules.get("torch.distributed")
    if dist is None or not dist.is_available() or not dist.is_initialized():
        return 1
    return dist.get_world_size(group)


def encode_batch(meshes, cfg=None, devices=None, group=None, device=None):
    """Every (already built, host-memory) mesh of a transcode job as ONE batch → list of `.drc` blobs in mesh order (None on the ranks
    that are not the destination of a sharded job).  torch.distributed initialised with more than one rank: the batch is dealt over
    the ranks by triangle count and gathered on rank 0 (RCCL for an nccl group).  Otherwise `devices` (a count, or "all") spreads it
    over the GPUs of this process (dmi_shard_meshes + dmi_meshes_prepare_devices + dmi_jobs_encode_devices); default: one GPU.
    (A pipelined form — dmi_meshes_prepare of stage k+1 beside dmi_jobs_encode of stage k — was 1.6 × slower on 256 meshes and is gone.)"""
    if not meshes:
        return []
    world = _world_size(group)
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


def transcode_files(sources, cfg=None, devices=None, group=None, device=None, pipeline=True, timings=None, copy=False, gather="files"):
    """BASELINE configs[3]: a LIST of glTF assets (GLB bytes, `.glb` / `.gltf` paths) → their Draco-compressed GLBs
    (io/gltf/transcoder.rs:134-151 runs the files one by one, io/gltf/encode.rs:1827-1842 their primitives one by one; here the triangle
    primitives of ALL files go through the device together: encode_raw_batch).  One GPU, `devices` GPUs of this process (each takes a
    share by triangle count, a thread per device), or the ranks of a torch.distributed job: the primitives are dealt by the triangle
    counts the JSON states, each rank builds and encodes ONLY its share, rank 0 gathers the blobs and reassembles the files.
    Returns [(glb_bytes, [blob, ...]), ...] in input order — on the destination rank; None on the other ranks of a sharded job.  On one device of one
    process the files are memoryviews of the library's output arena and the blobs memoryviews INTO their file (they compare equal to bytes; `bytes(blob)` copies one
    out; json.loads / pickle / dict keys want bytes; ONE surviving view keeps every 32 MiB arena block of the call alive) — copy=True returns bytes objects instead.
    timings (optional dict): parse_s (JSON), views_s (accessor views), build_s, prepare_s, encode_s, assemble_s, primitives_built (this rank).
    gather (ranks of a torch.distributed job only): "files" — the finished files travel to rank 0 (the default, what the paragraph above describes); "manifest" — the
    files STAY on the rank that made them (a transcoder's outputs are files: each rank writes its own) and only a manifest crosses the ranks: EVERY rank gets the
    list in input order with (glb, blobs) at the indices it owns and None elsewhere, and timings["manifest"] = (sizes, xxh64 digests) of all files, uint64 arrays."""
    import gc
    import time
    # The call allocates a few hundred thousand small objects (JSON trees, views, jobs) while four stage threads share the interpreter: a full
    # collection triggered in the middle stops all of them.  Collection is paused for the duration of the call (restored on the way out).
    gc_was = gc.isenabled() and os.environ.get("DMI_TRANSCODE_GC", "0") == "0"
    if gc_was:
        gc.disable()
    try:
        out = _transcode_files(sources, cfg, devices, group, device, pipeline, timings, gather)
        if copy and out is not None:
            # bytes objects of the caller's own (picklable, hashable, json-loadable) instead of views that keep the call's arena blocks alive
            out = [None if e is None else (bytes(e[0]), [bytes(b) for b in e[1]]) for e in out]
        return out
    finally:
        if gc_was:
            gc.enable()


def _native_assets(sources):
    """The sources of a transcode as dmi_transcode_assets takes them: GLB bytes as they are, a `.gltf` as (JSON text, buffers) with its buffers resolved here
    (a binding.AssetList — marshalled ahead of time — passes through)."""
    from .binding import AssetList
    if isinstance(sources, AssetList):
        return sources
    assets = []
    for src in sources:
        if isinstance(src, (bytes, bytearray, memoryview)):
            assets.append(src)
            continue
        data = open(src, "rb").read()
        if data[:4] == b"glTF":
            assets.append(data)
        else:
            _, buffers = load_document(src)
            assets.append((data, buffers))
    return assets


def _transcode_files(sources, cfg, devices, group, device, pipeline, timings, gather="files"):
    import time
    tm = timings if timings is not None else {}
    t0 = time.perf_counter()
    world0 = _world_size(group)
    if world0 == 1 and pipeline and os.environ.get("DMI_TRANSCODE_PYTHON", "0") == "0" and sources:
        # One process: the whole loop — container + JSON parse, primitive plans, accessor descriptors, stages on one dmi_transcoder per device,
        # file assembly — runs inside the library (dmi_transcode_assets, csrc/dmi_gltf.cpp); the files come back as views of its memory.
        # The steps below are the same loop in Python: the tests hold the two against each other, ranks of a torch.distributed job use it.
        n_dev = device_count() if devices == "all" else int(devices or 1)
        n_dev = max(1, min(n_dev, device_count()))
        base = cfg.device if cfg is not None else 0
        try:
            out, st = transcode_assets(_native_assets(sources), cfg, devices=list(range(n_dev)) if n_dev > 1 else [base])
        except DracoMiError as e:
            if "gltf:" in str(e):
                raise ValueError(str(e)) from None
            raise
        tm.update({"parse_s": st["parse_ms"] * 1e-3, "views_s": 0.0, "build_s": st["build_ms"] * 1e-3, "prepare_s": st["prepare_ms"] * 1e-3,
                   "encode_s": st["encode_ms"] * 1e-3, "assemble_s": st["assemble_ms"] * 1e-3, "primitives_built": st["primitives"], "native": st})
        return out
    if world0 > 1 and pipeline and os.environ.get("DMI_TRANSCODE_PYTHON", "0") == "0" and sources and encode_raw_batch is _ENCODE_RAW_BATCH:
        return _transcode_files_ranks(sources, cfg, device, group, tm, gather)
    if gather != "files" and world0 > 1:
        raise ValueError("gather='manifest' needs the library's own transcode loop (pipeline=True)")
    docs = [load_document(src) for src in sources]
    per_file = [_plan(doc) for doc, _ in docs]
    flat = [(fi, pi) for fi, prims in enumerate(per_file) for pi in range(len(prims))]
    weights = [per_file[fi][pi][2] for fi, pi in flat]
    world = _world_size(group)
    rank = sys.modules["torch.distributed"].get_rank(group) if world > 1 else 0
    if world > 1:
        from . import distributed
        mine = distributed.shard_indices(len(flat), rank, world, weights=weights)
        cfg = distributed._rank_config(cfg, device)
    else:
        mine = list(range(len(flat)))
    built = [0]

    def raw_of(k):   # the k-th primitive of this rank's share (made when its stage is built)
        fi, pi = flat[mine[k]]
        built[0] += 1
        return primitive_to_raw(docs[fi][0], docs[fi][1], per_file[fi][pi][0])[0]

    w_mine = [weights[i] for i in mine]
    tm["parse_s"] = time.perf_counter() - t0
    n_dev = device_count() if devices == "all" else int(devices or 1)
    n_dev = max(1, min(n_dev, device_count()))
    if world == 1 and n_dev > 1 and mine:
        from .distributed import shard_indices
        local = [None] * len(mine)
        errs = []

        def run(d):
            try:
                idx = shard_indices(len(mine), d, n_dev, weights=w_mine)
                c = Config(**{**(cfg.__dict__ if cfg else {}), "device": d})
                for i, r in zip(idx, encode_raw_batch(lambda k: raw_of(idx[k]), c, pipeline=pipeline, weights=[w_mine[i] for i in idx])):
                    local[i] = r
            except BaseException as e:                                    # noqa: BLE001
                errs.append(e)

        threads = [threading.Thread(target=run, args=(d,)) for d in range(n_dev)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errs:
            raise errs[0]
    else:
        # one device: a file is reassembled as soon as its last primitive is coded (a fourth stage beside the device work of the next ones)
        assembled = [None] * len(docs)
        left = [len(prims) for prims in per_file]
        first = [0] * len(docs)
        for fi in range(1, len(docs)):
            first[fi] = first[fi - 1] + len(per_file[fi - 1])
        tm["assemble_s"] = 0.0

        def on_done(indices, out):
            ta = time.perf_counter()
            for k in indices:
                fi = flat[mine[k]][0]
                left[fi] -= 1
                if left[fi] == 0 and world == 1:
                    res = out[first[fi]: first[fi] + len(per_file[fi])]
                    assembled[fi] = _assemble(docs[fi][0], docs[fi][1], per_file[fi], res)
            tm["assemble_s"] += time.perf_counter() - ta

        keep = [] if world == 1 else None                                  # the stages' library-owned outputs: views of them go straight into the files
        if world > 1 and pipeline and mine and encode_raw_batch is _ENCODE_RAW_BATCH and os.environ.get("DMI_TRANSCODE_PYTHON", "0") == "0":
            local = _share_native(raw_of, w_mine, cfg, tm)                 # (a rank's share through the library's stage loop; blobs as bytes: they travel)
        else:
            try:
                local = encode_raw_batch(raw_of, cfg, pipeline=pipeline, timings=tm, weights=w_mine, on_done=on_done if world == 1 else None, keep=keep)
            finally:
                for raw in keep or []:
                    raw.free()
        if world == 1:
            tm["primitives_built"] = built[0]
            for fi in range(len(docs)):
                if assembled[fi] is None:   # (a file without a compressible primitive)
                    assembled[fi] = _assemble(docs[fi][0], docs[fi][1], per_file[fi], [])
            return assembled
    tm["primitives_built"] = built[0]
    if world > 1:
        # the blobs travel with their face / point counts (rank 0 writes them into the placeholder accessors); 8 zero bytes = no face left
        payloads = [(np.array([r[1], r[2]], np.uint32).tobytes() + r[0]) if r is not None else b"\0" * 8 for r in local]
        got = distributed.gather_blob_lists(payloads, mine, len(flat), device=device, group=group)
        if got is None:
            return None
        results = []
        for b in got:
            b = bytes(b)
            nf, npts = np.frombuffer(b[:8], np.uint32)
            results.append((b[8:], int(nf), int(npts)) if len(b) > 8 else None)
    else:
        results = local
    t1 = time.perf_counter()
    out, at = [], 0
    for (doc, buffers), prims in zip(docs, per_file):
        mine_r = results[at: at + len(prims)]
        at += len(prims)
        glb, views = _assemble(doc, buffers, prims, mine_r)
        out.append((glb, [bytes(v) for v in views]))                            # (bytes: these results travel between processes / threads)
    tm["assemble_s"] = time.perf_counter() - t1
    return out


def _source_bytes(src):
    """What stands for a file's triangle count before anything is parsed: its size in bytes (a `.gltf`: the document + whatever lies beside it under its stem)."""
    if isinstance(src, (bytes, bytearray, memoryview)):
        return len(src)
    n = os.path.getsize(src)
    stem = os.path.splitext(src)[0]
    for ext in (".bin", "0.bin"):
        if os.path.exists(stem + ext):
            n += os.path.getsize(stem + ext)
    return n


_TRANSCODE_ASSETS = None   # (tests put a stand-in here: the ranks' control flow without a device)


def _transcode_files_ranks(sources, cfg, device, group, tm, gather="files"):
    """The rank-sharded form (round 6): the FILES are dealt to the ranks by their size in bytes (LPT) before anything is parsed — no rank reads, parses or
    plans a document it does not own —, every rank runs the library's own loop over its files (dmi_transcode_assets: parse pool, stage pipeline, assembly) and
    the FINISHED files travel: one size exchange + one gather of [blob table | GLB] payloads onto rank 0, no reassembly there.  (Round 5 parsed every JSON
    on every rank, dealt primitives, gathered blobs padded to the largest share and rebuilt every file on rank 0.)"""
    import time
    import torch.distributed as dist
    from . import distributed
    t0 = time.perf_counter()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [_source_bytes(src) for src in sources]
    mine = distributed.shard_indices(len(sources), rank, world, weights=sizes)
    cfg = distributed._rank_config(cfg, device)
    run = _TRANSCODE_ASSETS or transcode_assets
    own = [sources[i] for i in mine]
    if own:
        out, st = run(_native_assets(own), cfg, devices=[cfg.device])
    else:
        out, st = [], {"parse_ms": 0.0, "build_ms": 0.0, "prepare_ms": 0.0, "encode_ms": 0.0, "assemble_ms": 0.0, "primitives": 0}
    tm.update({"parse_s": st["parse_ms"] * 1e-3, "views_s": 0.0, "build_s": st["build_ms"] * 1e-3, "prepare_s": st["prepare_ms"] * 1e-3, "encode_s": st["encode_ms"] * 1e-3,
               "assemble_s": st["assemble_ms"] * 1e-3, "primitives_built": st["primitives"], "files_owned": len(own), "native": st, "transcode_s": time.perf_counter() - t0})
    if gather == "manifest":
        # the files stay here; what crosses the ranks is 16 bytes per file: size and xxh64 digest, every rank's entries summed into one array (an entry has one owner)
        import torch
        import xxhash
        t1 = time.perf_counter()
        man = np.zeros(2 * len(sources), np.int64)
        for i, (glb, _) in zip(mine, out):
            man[2 * i] = len(glb)
            man[2 * i + 1] = np.uint64(xxhash.xxh64(glb).intdigest()).astype(np.int64)
        dev = device if device is not None else torch.device("cpu")
        t = torch.from_numpy(man).to(dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        man = t.cpu().numpy().view(np.uint64)
        tm["manifest"] = (man[0::2].copy(), man[1::2].copy())
        tm["gather_s"] = time.perf_counter() - t1
        results = [None] * len(sources)
        for i, e in zip(mine, out):
            results[i] = e
        return results
    if gather != "files":
        raise ValueError("gather: 'files' or 'manifest'")
    payloads = []
    for glb, blobs in out:
        # where the blobs lie inside their file: [count | (offset, length) …] in front of the file's bytes
        g = np.frombuffer(glb, np.uint8)
        base = g.ctypes.data if len(g) else 0
        tab = np.empty(1 + 2 * len(blobs), np.uint64)
        tab[0] = len(blobs)
        for k, b in enumerate(blobs):
            v = np.frombuffer(b, np.uint8)
            tab[1 + 2 * k], tab[2 + 2 * k] = (v.ctypes.data - base if len(v) else 0), len(v)
        payloads.append(tab.tobytes() + bytes(glb))
    t1 = time.perf_counter()
    got = distributed.gather_blob_lists(payloads, mine, len(sources), device=device, group=group)
    tm["gather_s"] = time.perf_counter() - t1
    if got is None:
        return None
    results = []
    for p in got:
        p = bytes(p)
        n = int(np.frombuffer(p, np.uint64, count=1)[0])
        tab = np.frombuffer(p, np.uint64, count=1 + 2 * n)
        glb = p[(1 + 2 * n) * 8:]
        results.append((glb, [glb[int(tab[1 + 2 * k]): int(tab[1 + 2 * k]) + int(tab[2 + 2 * k])] for k in range(n)]))
    return results


def _share_native(raw_of, w_mine, cfg, tm):
    """A rank's share of a sharded transcode through dmi_transcoder → [(blob bytes, num_faces, num_points) or None] in the share's order."""
    import time
    from .binding import Transcoder
    n = len(w_mine)
    tm["views_s"] = 0.0
    with Transcoder(cfg, sum(w_mine), n, stage_triangles=PIPELINE_TRIANGLES) as t:
        slice_tris = max(1, sum(w_mine) // 64)
        lo = 0
        while lo < n:
            t0 = time.perf_counter()
            hi, acc = lo, 0
            while hi < n and (hi == lo or acc + w_mine[hi] <= slice_tris):
                acc += w_mine[hi]
                hi += 1
            raws = [raw_of(k) for k in range(lo, hi)]
            tm["views_s"] += time.perf_counter() - t0
            t.push(raws)
            lo = hi
        t.finish()
        tm.update(t.timings())
        out = []
        for k in range(n):
            r = t.result(k)
            out.append(None if r is None else (r[0][0].tobytes() + r[0][1].tobytes(), r[1], r[2]))
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
