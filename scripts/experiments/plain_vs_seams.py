import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import draco_oxide_amd as dmi
from draco_oxide_amd import synth
n_meshes = 256
rng = np.random.default_rng(synth.SEED)
tris = np.exp(rng.uniform(np.log(2e3), np.log(2e5), size=n_meshes))
for seams in (False, True):
    raws, total = [], 0
    for k, t in enumerate(tris):
        n = max(8, synth.grid_size_for_triangles(t))
        if seams:
            faces, pos, nrm, uv = synth.seam_torus_rows(n, seed=synth.SEED + 7 * k)
        else:
            faces, pos, nrm, uv = synth.torus_grid(n, synth.SEED + 7 * k, True, True, False)
        rm = dmi.RawMesh()
        rm.add_attribute(pos, dmi.ATT_POSITION, dmi.DOMAIN_POSITION)
        rm.add_attribute(nrm, dmi.ATT_NORMAL, dmi.DOMAIN_CORNER, [0])
        rm.add_attribute(uv, dmi.ATT_TEXCOORD, dmi.DOMAIN_CORNER, [0])
        rm.set_indices(faces.ravel())
        raws.append(rm); total += len(faces)
    dmi.init(0)
    for rep in range(4):
        t0 = time.perf_counter(); batch = dmi.meshes_build(raws); t1 = time.perf_counter()
        jobs = dmi.built_meshes_prepare(batch); t2 = time.perf_counter()
        with dmi.jobs_encode_raw(jobs) as out: nbytes = out.nbytes
        t3 = time.perf_counter()
        for j in jobs: j.close()
        batch.free()
        print(f"seams={seams}: {total} triangles: build {(t1-t0)*1e3:.2f} ms, built_prepare {(t2-t1)*1e3:.2f}, encode {(t3-t2)*1e3:.2f}; total {(t3-t0)*1e3:.2f} ms = {total/(t3-t0)/1e6:.1f} Mtri/s", flush=True)
