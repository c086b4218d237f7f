"""dmi_meshes_prepare of the batch workload, a few times (DMI_TRACE=1 prints the stage split of the device form)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import draco_oxide_amd as dmi
from draco_oxide_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
meshes = synth.batch_meshes(n)
total = sum(len(m.faces) for m in meshes)
for r in range(reps):
    t0 = time.perf_counter()
    jobs = dmi.meshes_prepare(meshes, dmi.Config())
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    with dmi.jobs_encode_raw(jobs) as b:
        pass
    de = time.perf_counter() - t1
    for j in jobs:
        j.close()
    print(f"prepare {n} meshes ({total} triangles): {dt * 1e3:.1f} ms, encode {de * 1e3:.1f} ms, end to end {total / (dt + de) / 1e6:.1f} Mtri/s", flush=True)
