"""Batch regime (BASELINE configs[3] shape): many independent meshes, F log-uniform in [2k, 200k],
pos+nrm+uv, all jobs resident, ONE dmi_jobs_encode per step.  Prints aggregate Mtri/s."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import draco_oxide_amd as dmi
from draco_oxide_amd import synth

n_meshes = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
cfg = dmi.Config()
meshes = synth.batch_meshes(n_meshes)
total = sum(len(m.faces) for m in meshes)
t0 = time.time()
jobs = dmi.meshes_prepare(meshes, cfg)   # corner tables, Edgebreaker, sequencers, uploads: thread pool inside the library
prep = time.time() - t0
outs = dmi.jobs_encode(jobs)   # warm-up
torch.cuda.synchronize()
# the product boundary: one dmi_jobs_encode call (outputs in library-owned buffers) + dmi_free of every buffer
t0 = time.perf_counter()
for _ in range(steps):
    with dmi.jobs_encode_raw(jobs) as batch:
        pass
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
# the same through the Python convenience wrapper, which also copies every section into a `bytes` object
t0 = time.perf_counter()
for _ in range(steps):
    outs = dmi.jobs_encode(jobs)
dt_py = (time.perf_counter() - t0) / steps
with dmi.jobs_encode_raw(jobs) as batch:
    assert batch.nbytes == sum(len(o) for o in outs)
# spot-check against single-job encodes
for j in (0, len(jobs) // 2, len(jobs) - 1):
    assert outs[j] == jobs[j].encode()
print(json.dumps({"workload": f"batch of {n_meshes} meshes, F log-uniform [2k,200k], pos+nrm+uv", "triangles": int(total), "ms_per_batch": round(dt * 1e3, 3),
                  "mtri_per_s": round(total / dt / 1e6, 2), "ms_per_batch_python_bytes": round(dt_py * 1e3, 3), "bytes": int(sum(len(o) for o in outs)), "host_prepare_s": round(prep, 2)}))
