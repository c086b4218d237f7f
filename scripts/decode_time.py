"""Decoder timing (needs an MI355X): a .drc the library wrote, read back by dmi_decode_mesh; stage split from dmi_last_decode_timings.
usage: decode_time.py [grid_n ...]   (grid_n 2237 = 10M triangles)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import draco_oxide_amd as dmi
from draco_oxide_amd import synth

dmi.init()
for n in [int(x) for x in sys.argv[1:]] or [708, 2237]:
    mesh = synth.torus_mesh(n)
    drc = dmi.encode_mesh(mesh)
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        dec = dmi.decode_mesh(drc)
        dt = time.perf_counter() - t0
        t = dmi.last_decode_timings()
        if best is None or t["call_ms"] < best["call_ms"]:
            best = dict(t, python_call_ms=dt * 1e3)
    assert dec["faces"].shape[0] == len(mesh.faces)
    print(json.dumps(dict(triangles=len(mesh.faces), drc_bytes=len(drc), mtri_per_s=len(mesh.faces) / best["call_ms"] / 1e3, **{k: round(v, 2) for k, v in best.items()})))
