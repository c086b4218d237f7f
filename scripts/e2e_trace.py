#!/usr/bin/env python3
"""One 10M-triangle mesh through dmi_encode_mesh (host memory) and dmi_encode_mesh_device (mesh in HBM) with DMI_TRACE=1: where the
end-to-end time goes."""
import os, sys, time
os.environ["DMI_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import draco_oxide_amd as dmi
mesh = dmi.synth.torus_mesh(int(sys.argv[1]) if len(sys.argv) > 1 else 2236)
def thp():
    for l in open("/proc/self/smaps_rollup"):
        if "AnonHuge" in l: return l.split()[1] + " kB huge"
for k in range(4):
    sys.stderr.write(f"---- host memory, run {k}\n"); sys.stderr.flush()
    t = time.perf_counter(); drc = dmi.encode_mesh(mesh); dt = time.perf_counter() - t
    sys.stderr.write(f"encode_mesh {dt * 1e3:.1f} ms, {len(drc)} bytes, {thp()}\n"); sys.stderr.flush()
dm = dmi.DeviceMesh.upload(mesh)
for k in range(4):
    sys.stderr.write(f"---- mesh in HBM, run {k}\n"); sys.stderr.flush()
    t = time.perf_counter(); drc2 = dmi.encode_mesh_device(dm); dt = time.perf_counter() - t
    sys.stderr.write(f"encode_mesh_device {dt * 1e3:.1f} ms, same bytes: {drc2 == drc}, {thp()}\n"); sys.stderr.flush()
