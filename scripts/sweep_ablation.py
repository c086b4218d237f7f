#!/usr/bin/env python3
"""Where do the vector slots of the fused predictor sweep go?  The 10M-triangle workload with attribute subsets: each subset runs its own
instantiation of the sweep (pnu / pn / pu / positions only), so the differences are the cost of the normal and the texcoord tiers."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import draco_oxide_amd as dmi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
subsets = ((True, True), (True, False), (False, True), (False, False))
if len(sys.argv) > 2:
    subsets = subsets[: int(sys.argv[2])]
for normals, uvs in subsets:
    mesh = dmi.synth.torus_mesh(n, normals=normals, uvs=uvs)
    job = dmi.mesh_prepare(mesh, dmi.Config(flags=dmi.FLAG_TIMINGS))
    best = None
    for _ in range(6):
        job.encode_raw().free()
        t = job.timings()
        if best is None or t["predict_ms"] < best["predict_ms"]:
            best = t
    print(f"normals={int(normals)} uvs={int(uvs)}  quantize {best['quantize_ms']:.3f} ms  predict {best['predict_ms']:.3f} ms", flush=True)
    job.close()
