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
