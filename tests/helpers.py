"""Shared test plumbing: oracle session ↔ product Mesh conversion."""
import os

import numpy as np

import draco_oxide_amd as dmi
import orc

DATA = os.path.join(os.path.dirname(__file__), "golden", "data")


def product_mesh_from_oracle(sess):
    """The oracle's built mesh (unique values + p2v) expressed as a product `Mesh`."""
    atts = sess.attributes()
    ids = {a["id"]: i for i, a in enumerate(atts)}
    out = []
    for a in atts:
        parent = ids[a["parents"][0]] if a["parents"] else -1
        out.append(dmi.Attribute(a["data"], a["type"], a["domain"], unique_id=a["id"], parent_index=parent, point_to_value=a["p2v"], num_points=a["len"]))
    return dmi.Mesh(sess.faces(), out)


def oracle_from_product_mesh(mesh):
    """Feed a product Mesh (already built) to the oracle through its builder (dedup is then the identity
    for unique values; p2v maps are expanded back to per-point rows)."""
    specs = []
    for a in mesh.attributes:
        rows = a.values if a.point_to_value is None else a.values[a.point_to_value]
        parents = [] if a.parent_index < 0 else [mesh.attributes[a.parent_index].unique_id]
        specs.append(dict(data=rows, type=a.att_type, domain=a.domain, parents=parents))
    return orc.Session.from_arrays(mesh.faces, specs)


def obj_session(name, faithful=False):
    return orc.Session.from_obj(os.path.join(DATA, name + ".obj"), faithful=faithful)


def tables_from_oracle(sess, n_atts):
    """dmi_corner_table dicts (one per attribute) from the oracle's dumps (after sess.encode())."""
    c2p = sess.blob("ct.c2p", np.uint32)
    tabs = []
    for i in range(n_atts):
        if i == 0 or len(sess.blob(f"at{i-1}.c2v", np.uint32)) == 0:
            t = dict(corner_to_point=c2p, corner_to_vertex=sess.blob("ct.c2v", np.uint32), opposite=sess.blob("ct.opp", np.uint32),
                     left_most_corner=sess.blob("ct.lmc", np.uint32))
        else:
            t = dict(corner_to_point=c2p, corner_to_vertex=sess.blob(f"at{i-1}.c2v", np.uint32), opposite=sess.blob(f"at{i-1}.opp", np.uint32),
                     left_most_corner=sess.blob(f"at{i-1}.lmc", np.uint32))
        t["num_vertices"] = len(t["left_most_corner"])
        t["sequence"] = sess.blob(f"att{i}.seq", np.uint32)
        tabs.append(t)
    return tabs


def oracle_values_by_point(table, universal_table, decoded, num_points):
    """The oracle decoder's values (one per sequence entry = per vertex of the attribute's table) laid out per POINT the way the product
    lays them out: a point takes the value of the vertex of its LAST corner (a point can be met by two vertices of a table — a
    non-manifold vertex split in two — whose decoded values differ when the reference's lossy normal case hits one of them).
    Returns (values [num_points, n], referenced [num_points] bool)."""
    c2p = np.asarray(universal_table["corner_to_point"])
    c2v = np.asarray(table["corner_to_vertex"])
    seq = np.asarray(table["sequence"])
    vals = np.asarray(decoded["values"])
    by_vertex = np.zeros((int(table["num_vertices"]), vals.shape[1]), vals.dtype)
    by_vertex[c2v[seq]] = vals
    last = np.full(num_points, -1, np.int64)
    last[c2p] = np.arange(len(c2p))                     # (numpy keeps the last assignment of a repeated index: the largest corner)
    out = np.zeros((num_points, vals.shape[1]), vals.dtype)
    ref = last >= 0
    out[ref] = by_vertex[c2v[last[ref]]]
    return out, ref


def nan_twin_primitives():
    """Primitives that hit the one place where the reference's point merge is not "same value ids": core/mesh/builder.rs:254-279 hashes
    the BYTES of each attribute's unique value at the point.  A row holding a NaN equals nothing in the value dedup
    (core/attribute/mod.rs:394-452) and keeps a value of its own, but two byte-identical NaN rows hash equal — their points merge when
    every other attribute agrees.  → [(name, specs [(rows, type, domain, parents)], faces, points expected after the build)]."""
    nan = np.float32(np.nan)

    def base():
        pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [2, 0, 0], [2, 1, 0], [0, 2, 0], [1, 2, 0]], np.float32)
        nrm = np.tile(np.array([[0, 0, 1]], np.float32), (8, 1))
        uv = (pos[:, :2] / np.float32(2.0)).astype(np.float32)
        faces = np.array([[0, 1, 2], [1, 3, 2], [1, 4, 3], [4, 5, 3], [2, 3, 6], [3, 7, 6]], np.uint32)
        return pos, nrm, uv, faces

    def specs(pos, nrm, uv):
        return [(pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION, []), (nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, [0]), (uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, [0])]

    out = []
    # 1. NaN in positions, the two points byte-identical in every attribute: they merge (7 points), the second NaN value leaves the buffer
    pos, nrm, uv, faces = base()
    pos[5] = [nan, 1, 0]; pos[7] = pos[5]; nrm[7] = nrm[5]; uv[7] = uv[5]
    out.append(("nan_positions_twins", specs(pos, nrm, uv), faces, 7))
    # 2. NaN in normals only, twins
    pos, nrm, uv, faces = base()
    pos[7] = pos[5]; uv[7] = uv[5]; nrm[5] = [0, nan, 1]; nrm[7] = nrm[5]
    out.append(("nan_normals_twins", specs(pos, nrm, uv), faces, 7))
    # 3. NaN rows whose payloads differ: different bytes, no merge
    pos, nrm, uv, faces = base()
    pos[7] = pos[5]; uv[7] = uv[5]; nrm[5] = [0, nan, 1]; nrm[7] = nrm[5]
    nrm.view(np.uint32)[7, 1] ^= 1
    out.append(("nan_payloads_differ", specs(pos, nrm, uv), faces, 8))
    # 4. NaN rows that differ in the sign of a zero elsewhere in the row: bytes differ, no merge (a NaN-free row WOULD merge: case 6)
    pos, nrm, uv, faces = base()
    pos[7] = pos[5]; uv[7] = uv[5]; nrm[5] = [0.0, nan, 1]; nrm[7] = [-0.0, nan, 1]
    out.append(("nan_rows_signed_zero", specs(pos, nrm, uv), faces, 8))
    # 5. byte-identical NaN rows in one attribute, another attribute differs: no merge, both NaN values stay
    pos, nrm, uv, faces = base()
    nrm[5] = [0, nan, 1]; nrm[7] = nrm[5]
    out.append(("nan_twins_other_attribute_differs", specs(pos, nrm, uv), faces, 8))
    # 6. -0.0 first, 0.0 later (and the other way round in another attribute): `==` rows, ONE value carrying the first occurrence's bytes: merge
    pos, nrm, uv, faces = base()
    pos[5] = [2, 1, -0.0]; pos[7] = [2, 1, 0.0]; nrm[5] = [0.0, 0, 1]; nrm[7] = [-0.0, 0, 1]; uv[7] = uv[5]
    out.append(("signed_zero_first_occurrences", specs(pos, nrm, uv), faces, 7))
    # 7. three byte-identical NaN points + a NaN row referenced by no face + NaN twins in TWO attributes
    pos, nrm, uv, faces = base()
    pos = np.vstack([pos, [[nan, nan, nan]], [[9, 9, 9]]]).astype(np.float32)
    nrm = np.vstack([nrm, [[0, 0, 1]], [[0, 0, 1]]]).astype(np.float32)
    uv = np.vstack([uv, [[0, 0]], [[nan, 0]]]).astype(np.float32)
    for k in (5, 7, 8):
        pos[k] = [nan, nan, nan]; nrm[k] = [nan, 0, 1]; uv[k] = [0.5, nan]
    faces = np.vstack([faces, [[0, 8, 2]]]).astype(np.uint32)
    out.append(("nan_triplet_two_attributes", specs(pos, nrm, uv), faces, 7))   # 10 points: 5, 7, 8 become one, point 9 is referenced by no face
    # 8. NaN twins where the attribute has NO other duplicate (no point → value map before the merge)
    pos, nrm, uv, faces = base()
    nrm = (pos + np.float32(0.5)).astype(np.float32)
    pos[5] = [nan, 1, 0]; pos[7] = pos[5]; nrm[7] = nrm[5]; uv[7] = uv[5]
    out.append(("nan_twins_without_a_map", specs(pos, nrm, uv), faces, 7))
    return out


def oracle_session_of_specs(specs, faces):
    return orc.Session.from_arrays(faces, [dict(data=np.ascontiguousarray(r), type=t, domain=d, parents=list(p)) for r, t, d, p in specs])


def irregular_grid_mesh(n, seed, open_boundary=False, shuffle_points=False, normals=True, uvs=True, hubs=40, single_splits=200):
    """A torus grid (synth.torus_grid: ≥ 2^16 faces from n = 182 on) made irregular while staying manifold, per-point attributes throughout (no
    point → value map: the layout a whole-mesh call's early stage takes): `hubs` vertices get EVERY incident face split in three around a new
    centre vertex — the hub's valence doubles to 12 (its fan row overflows: the corner-table walk), its neighbours' grows to 8, the centres have
    valence 3 —, `single_splits` further faces are split on their own (valence 7 / 3), optionally the grid is open (boundary fans, vertices
    without a parallelogram) and the point order is scrambled (the value-order records of the early stage are then read at random)."""
    from draco_oxide_amd import synth
    rng = np.random.default_rng(seed)
    faces, pos, nrm, uv = synth.torus_grid(n, seed=seed, normals=normals, uvs=uvs, open_boundary=open_boundary)
    faces = faces.astype(np.int64)
    F, V = len(faces), len(pos)
    # faces around each chosen hub (hubs far apart: no face is split twice)
    hub_ids = rng.choice(V, size=hubs * 4, replace=False)
    taken = np.zeros(F, bool)
    split = []
    inc = {}
    for v in hub_ids:
        fs = np.nonzero((faces == v).any(axis=1))[0]
        if len(fs) == 0 or taken[fs].any():
            continue
        nb = np.unique(faces[fs])
        around = np.nonzero(np.isin(faces, nb).any(axis=1))[0]
        if taken[around].any():
            continue
        taken[around] = True
        split.extend(fs.tolist())
        if len(split) >= hubs * 6:
            break
    free = np.nonzero(~taken)[0]
    split.extend(rng.choice(free, size=min(single_splits, len(free)), replace=False).tolist())
    split = np.asarray(sorted(set(split)), np.int64)
    centre = V + np.arange(len(split))
    a, b, c = faces[split, 0], faces[split, 1], faces[split, 2]
    w = rng.uniform(0.2, 0.5, size=(len(split), 3))
    w /= w.sum(axis=1, keepdims=True)

    def mix(arr):
        return (arr[a] * w[:, :1] + arr[b] * w[:, 1:2] + arr[c] * w[:, 2:3]).astype(np.float32)
    pos = np.concatenate([pos, mix(pos) + rng.uniform(-1e-4, 1e-4, size=(len(split), 3)).astype(np.float32)])
    if nrm is not None:
        m = mix(nrm) + rng.uniform(-1e-3, 1e-3, size=(len(split), 3)).astype(np.float32)
        nrm = np.concatenate([nrm, (m / np.linalg.norm(m, axis=1, keepdims=True)).astype(np.float32)])
    if uv is not None:
        uv = np.concatenate([uv, np.clip(mix(uv) + rng.uniform(-1e-4, 1e-4, size=(len(split), 2)), 0, 1).astype(np.float32)])
    keep = np.ones(F, bool)
    keep[split] = False
    new = np.concatenate([np.stack([a, b, centre], 1), np.stack([b, c, centre], 1), np.stack([c, a, centre], 1)])
    faces = np.concatenate([faces[keep], new])
    faces = faces[rng.permutation(len(faces))]
    if shuffle_points:
        perm = rng.permutation(len(pos))            # new id of old point
        inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
        faces = perm[faces]
        pos = pos[inv]; nrm = None if nrm is None else nrm[inv]; uv = None if uv is None else uv[inv]
    atts = [dmi.Attribute(pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION, unique_id=0)]
    if nrm is not None:
        atts.append(dmi.Attribute(nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, unique_id=len(atts), parent_index=0))
    if uv is not None:
        atts.append(dmi.Attribute(uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, unique_id=len(atts), parent_index=0))
    for arr in (pos, nrm, uv):   # the early stage takes per-point attributes: no two rows may merge
        if arr is not None:
            rows = np.ascontiguousarray(arr + np.float32(0.0)).view(np.dtype((np.void, arr.dtype.itemsize * arr.shape[1]))).ravel()
            assert len(np.unique(rows)) == len(rows)
    return dmi.Mesh(faces.astype(np.uint32), atts)
