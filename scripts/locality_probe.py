"""Probe: how much do the gather-bound kernels depend on the input index order?  Encodes the workload
mesh as generated (row-major along b) and with the grid axes swapped (row-major along a)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import draco_oxide_amd as dmi
from draco_oxide_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
for swap in (False, True):
    faces, pos, nrm, uv = synth.torus_grid(n)
    if swap:
        perm = np.arange(n * n).reshape(n, n).T.ravel()          # new vertex k ← old vertex perm[k]
        inv = np.empty_like(perm); inv[perm] = np.arange(n * n)
        pos, nrm, uv = pos[perm], nrm[perm], uv[perm]
        faces = inv[faces].astype(np.uint32)
        fo = np.arange(len(faces)).reshape(n, n, 2).transpose(1, 0, 2).ravel()
        faces = np.ascontiguousarray(faces[fo])
    mesh = dmi.Mesh(faces, [dmi.Attribute(pos, dmi.ATT_POSITION), dmi.Attribute(nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, 1, 0), dmi.Attribute(uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, 2, 0)])
    job = dmi.mesh_prepare(mesh, dmi.Config(flags=dmi.FLAG_TIMINGS))
    job.encode(); job.encode()
    t = job.timings()
    print("swap", swap, {k: round(v, 3) for k, v in t.items() if k.endswith("_ms")}, flush=True)
    job.close()
