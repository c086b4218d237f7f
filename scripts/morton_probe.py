"""Would a space-filling-curve face order speed the serial walks up?  Permute the 10M-triangle workload's faces by the Morton code of their
centroids (numpy), run dmi_encode_mesh on both orders with DMI_TRACE=1 and compare the traversal / sequencer times."""
import os, sys, time
os.environ["DMI_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
import draco_oxide_amd as dmi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
mesh = dmi.synth.torus_mesh(n)
pos = mesh.attributes[0].values
c = pos[mesh.faces].mean(axis=1)
q = ((c - c.min(0)) / (c.max(0) - c.min(0) + 1e-9) * 1023).astype(np.uint64)
def spread(x):
    x = (x | (x << 16)) & 0x030000FF0000FF
    x = (x | (x << 8)) & 0x0300F00F00F00F
    x = (x | (x << 4)) & 0x030C30C30C30C3
    x = (x | (x << 2)) & 0x09249249249249
    return x
key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
order = np.argsort(key, kind="stable")
# vertices renumbered by first use in the new face order (locality of the per-vertex flags too)
f2 = mesh.faces[order]
first = np.full(len(pos), -1, np.int64)
flat = f2.ravel()
uniq, idx = np.unique(flat, return_index=True)
vorder = uniq[np.argsort(idx)]
newid = np.empty(len(pos), np.int64); newid[vorder] = np.arange(len(vorder))
atts = [dmi.Attribute(a.values[vorder], a.att_type, a.domain, a.unique_id, a.parent_index) for a in mesh.attributes]
mesh2 = dmi.Mesh(newid[f2].astype(np.uint32), atts)
mesh3 = dmi.Mesh(f2, mesh.attributes)   # faces permuted only
for name, m in (("row-major (workload)", mesh), ("Morton faces + vertices", mesh2), ("Morton faces only", mesh3)):
    for k in range(3):
        sys.stderr.write(f"---- {name}, run {k}\n"); sys.stderr.flush()
        t = time.perf_counter(); drc = dmi.encode_mesh(m); dt = time.perf_counter() - t
        sys.stderr.write(f"encode_mesh {dt * 1e3:.1f} ms, {len(drc)} bytes\n"); sys.stderr.flush()
