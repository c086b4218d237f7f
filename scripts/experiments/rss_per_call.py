"""Resident memory of the process after every whole-mesh call (a leak would show as steady growth): python scripts/experiments/rss_per_call.py [calls]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import draco_oxide_amd as dmi
from draco_oxide_amd import synth
mesh = synth.torus_mesh(2236)
dm = dmi.DeviceMesh.upload(mesh, 0); cm = dm._c()
cfg = dmi.Config(device=0)
def rss():
    for line in open("/proc/self/smaps_rollup"):
        if line.startswith("Rss"): return int(line.split()[1]) // 1024
def n173():
    n = 0; size = 0
    for line in open("/proc/self/smaps"):
        p = line.split()
        if p and p[0] == "Size:": size = int(p[1])
        if p and p[0] == "Rss:" and 170 * 1024 <= size <= 176 * 1024: n += 1
    return n
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    with dmi.encode_mesh_device_raw(dm, cfg, cm) as out: pass
    print(f"call {k}: rss {rss()} MB, mappings of ≈ 173 MB: {n173()}", flush=True)
