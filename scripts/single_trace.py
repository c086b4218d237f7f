#!/usr/bin/env python3
"""One whole-mesh call (dmi_encode_mesh_device, the bench's `value`) with the library's stage trace: python scripts/single_trace.py [grid=2236] [calls=4]"""
import os
import sys
import time

os.environ.setdefault("DMI_TRACE", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import draco_oxide_amd as dmi  # noqa: E402
from draco_oxide_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mesh = synth.torus_mesh(n)
dm = dmi.DeviceMesh.upload(mesh, 0)
for k in range(calls):
    t0 = time.perf_counter()
    out = dmi.encode_mesh_device(dm)
    print(f"call {k}: {(time.perf_counter() - t0) * 1e3:.1f} ms, {len(out)} bytes", file=sys.stderr, flush=True)
