#!/usr/bin/env python3
"""dmi_jobs_encode of the 256-mesh batch with and without the hybrid tail (DMI_BATCH_TAIL is read once per process: run twice)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import draco_oxide_amd as dmi  # noqa: E402
from draco_oxide_amd import synth  # noqa: E402

meshes = synth.batch_meshes(256)
total = sum(len(m.faces) for m in meshes)
ref = None
for rep in range(4):
    jobs = dmi.meshes_prepare(meshes)
    t0 = time.perf_counter()
    with dmi.jobs_encode_raw(jobs) as out:
        first = time.perf_counter() - t0
        blobs = [out[i] for i in range(len(jobs))]
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        with dmi.jobs_encode_raw(jobs):
            pass
        ts.append(time.perf_counter() - t0)
    for j in jobs:
        j.close()
    print(f"DMI_BATCH_TAIL={os.environ.get('DMI_BATCH_TAIL', 'default')}: first encode of fresh jobs {first * 1e3:.2f} ms, resident re-encode {min(ts) * 1e3:.2f} ms ({total / min(ts) / 1e6:.0f} Mtri/s)", flush=True)
    import hashlib
    h = hashlib.sha256(b"".join(blobs)).hexdigest()[:16]
    print("  digest", h)
