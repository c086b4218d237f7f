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
