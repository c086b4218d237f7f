#!/usr/bin/env python3
"""Times dmi_meshes_build (device MeshBuilder::build) — one 10M-triangle primitive and a batch — beside the host builder, and the
transcode chain build → built_prepare → encode against the host-mesh chain.  Usage: python scripts/build_time.py [grid] [n_batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import draco_oxide_amd as dmi  # noqa: E402
from draco_oxide_amd import synth  # noqa: E402


def raw_of_grid(n, seed=synth.SEED):
    faces, pos, nrm, uv = synth.torus_grid(n, seed)
    rm = dmi.RawMesh()
    rm.add_attribute(pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION)
    rm.add_attribute(nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, [0])
    rm.add_attribute(uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, [0])
    rm.set_indices(faces.ravel())
    return rm, len(faces)


def main():
    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
    n_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    dmi.init(0)
    rm, f = raw_of_grid(grid)
    for rep in range(4):
        t0 = time.perf_counter()
        batch = dmi.meshes_build([rm])
        dt = time.perf_counter() - t0
        tm = dmi.last_build_timings()
        print(f"single {f} triangles: meshes_build {dt * 1e3:.2f} ms ({f / dt / 1e6:.1f} Mtri/s): pack {tm['pack_ms']:.2f}, kernels {tm['kernels_ms']:.2f}, call {tm['call_ms']:.2f}; "
              f"up {tm['bytes_up'] / 1e6:.0f} MB, down {tm['bytes_down'] / 1e6:.0f} MB", flush=True)
        batch.free()
    if grid <= 1200:
        b = dmi.MeshBuilder()
        for r, t, d, par in rm.atts:
            b.add_attribute(r, t, d, parents=list(par))
        b.set_connectivity_attribute(rm.indices.reshape(-1, 3))
        t0 = time.perf_counter()
        b.build()
        print(f"host builder (dmi_mesh_build incl. the Python copy-out): {(time.perf_counter() - t0) * 1e3:.1f} ms")
    # batch
    rng = np.random.default_rng(synth.SEED)
    tris = np.exp(rng.uniform(np.log(2e3), np.log(2e5), size=n_batch))
    raws, total = [], 0
    for k, t in enumerate(tris):
        r, nf = raw_of_grid(max(8, synth.grid_size_for_triangles(t)), seed=synth.SEED + 7 * k)
        raws.append(r)
        total += nf
    for rep in range(4):
        t0 = time.perf_counter()
        batch = dmi.meshes_build(raws)
        t1 = time.perf_counter()
        tm = dmi.last_build_timings()
        jobs = dmi.built_meshes_prepare(batch)
        t2 = time.perf_counter()
        with dmi.jobs_encode_raw(jobs) as out:
            nbytes = out.nbytes
        t3 = time.perf_counter()
        for j in jobs:
            j.close()
        batch.free()
        print(f"batch {n_batch} meshes / {total} triangles: build {(t1 - t0) * 1e3:.2f} ms (pack {tm['pack_ms']:.2f}, kernels {tm['kernels_ms']:.2f}), built_prepare {(t2 - t1) * 1e3:.2f}, "
              f"encode {(t3 - t2) * 1e3:.2f}; total {(t3 - t0) * 1e3:.2f} ms = {total / (t3 - t0) / 1e6:.1f} Mtri/s ({nbytes} bytes)", flush=True)
    meshes = synth.batch_meshes(n_batch)
    for rep in range(3):
        t0 = time.perf_counter()
        jobs = dmi.meshes_prepare(meshes)
        t1 = time.perf_counter()
        with dmi.jobs_encode_raw(jobs) as out:
            pass
        t2 = time.perf_counter()
        for j in jobs:
            j.close()
        print(f"host-mesh chain: meshes_prepare {(t1 - t0) * 1e3:.2f} ms, encode {(t2 - t1) * 1e3:.2f}; total {total / (t2 - t0) / 1e6:.1f} Mtri/s", flush=True)


if __name__ == "__main__":
    main()
