#!/usr/bin/env python3
"""What one encode stage of the transcode pipeline spends where (≈ 6M triangles of batch meshes): python scripts/encode_stage_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import draco_oxide_amd as dmi  # noqa: E402
from draco_oxide_amd import synth  # noqa: E402

meshes = synth.batch_meshes(256)
acc, pick = 0, []
for m in meshes:
    if acc > (6 << 20):
        break
    pick.append(m)
    acc += len(m.faces)
print(len(pick), "meshes,", acc, "triangles")
for rep in range(4):
    jobs = dmi.meshes_prepare(pick)
    t0 = time.perf_counter()
    raw = dmi.jobs_encode_raw(jobs)
    t1 = time.perf_counter()
    sections = [raw[i] for i in range(len(jobs))]
    raw.free()
    t2 = time.perf_counter()
    blobs = [j.header_and_connectivity + s for j, s in zip(jobs, sections)]
    t3 = time.perf_counter()
    for j in jobs:
        j.close()
    t4 = time.perf_counter()
    print(f"dmi_jobs_encode {(t1 - t0) * 1e3:.2f} ms, copy sections out + free {(t2 - t1) * 1e3:.2f}, head + section {(t3 - t2) * 1e3:.2f}, close jobs {(t4 - t3) * 1e3:.2f}", flush=True)
