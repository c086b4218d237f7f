#!/usr/bin/env python3
"""dmi_meshes_prepare + dmi_jobs_encode of the 256-mesh batch with the library's stage trace: python scripts/prepare_trace.py [n_meshes=256] [calls=4]"""
import os
import sys
import time

os.environ.setdefault("DMI_TRACE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import draco_oxide_amd as dmi  # noqa: E402
from draco_oxide_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
meshes = synth.batch_meshes(n)
total = sum(len(m.faces) for m in meshes)
for k in range(calls):
    t0 = time.perf_counter()
    jobs = dmi.meshes_prepare(meshes)
    t1 = time.perf_counter()
    with dmi.jobs_encode_raw(jobs):
        pass
    t2 = time.perf_counter()
    for j in jobs:
        j.close()
    print(f"call {k}: prepare {(t1 - t0) * 1e3:.1f} ms, encode {(t2 - t1) * 1e3:.1f} ms, {total / (t2 - t0) / 1e6:.1f} Mtri/s", file=sys.stderr, flush=True)
