#!/usr/bin/env python3
"""One 10M-triangle mesh through dmi_encode_mesh with DMI_TRACE=1: where the end-to-end time goes."""
import os, sys, time
os.environ["DMI_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import draco_oxide_amd as dmi
mesh = dmi.synth.torus_mesh(int(sys.argv[1]) if len(sys.argv) > 1 else 2236)
for k in range(3):
    sys.stderr.write(f"---- run {k}\n"); sys.stderr.flush()
    t = time.perf_counter(); drc = dmi.encode_mesh(mesh); dt = time.perf_counter() - t
    sys.stderr.write(f"encode_mesh {dt * 1e3:.1f} ms, {len(drc)} bytes\n"); sys.stderr.flush()
