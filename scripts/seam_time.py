#!/usr/bin/env python3
"""Batch prepare + encode of meshes WITH UV seams (the shape real glTF assets have) beside the seam-free batch: python scripts/seam_time.py [n_meshes]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import draco_oxide_amd as dmi  # noqa: E402
from draco_oxide_amd import synth  # noqa: E402

n_meshes = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rng = np.random.default_rng(synth.SEED)
tris = np.exp(rng.uniform(np.log(2e3), np.log(2e5), size=n_meshes))
raws, total = [], 0
for k, t in enumerate(tris):
    faces, pos, nrm, uv = synth.seam_torus_rows(max(8, synth.grid_size_for_triangles(t)), seed=synth.SEED + 7 * k)
    rm = dmi.RawMesh()
    rm.add_attribute(pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION)
    rm.add_attribute(nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, [0])
    rm.add_attribute(uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, [0])
    rm.set_indices(faces.ravel())
    raws.append(rm)
    total += len(faces)
dmi.init(0)
for rep in range(4):
    t0 = time.perf_counter()
    batch = dmi.meshes_build(raws)
    t1 = time.perf_counter()
    jobs = dmi.built_meshes_prepare(batch)
    t2 = time.perf_counter()
    with dmi.jobs_encode_raw(jobs) as out:
        nbytes = out.nbytes
    t3 = time.perf_counter()
    for j in jobs:
        j.close()
    batch.free()
    print(f"seams: {n_meshes} meshes / {total} triangles: build {(t1 - t0) * 1e3:.2f} ms, built_prepare {(t2 - t1) * 1e3:.2f}, encode {(t3 - t2) * 1e3:.2f}; "
          f"total {(t3 - t0) * 1e3:.2f} ms = {total / (t3 - t0) / 1e6:.1f} Mtri/s ({nbytes} bytes)", flush=True)
# the same meshes as host meshes through dmi_meshes_prepare
with dmi.meshes_build(raws, host_values=True) as batch:
    meshes = [batch.mesh(j) for j in range(n_meshes)]
for rep in range(3):
    t0 = time.perf_counter()
    jobs = dmi.meshes_prepare(meshes)
    t1 = time.perf_counter()
    with dmi.jobs_encode_raw(jobs) as out:
        pass
    t2 = time.perf_counter()
    for j in jobs:
        j.close()
    print(f"seams, host meshes: meshes_prepare {(t1 - t0) * 1e3:.2f} ms, encode {(t2 - t1) * 1e3:.2f}; total {total / (t2 - t0) / 1e6:.1f} Mtri/s", flush=True)
