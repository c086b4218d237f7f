#!/usr/bin/env python3
"""Pipelined transcode of exporter-style GLBs with UV seams: python scripts/seam_transcode_time.py [n_files]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import draco_oxide_amd as dmi  # noqa: E402
from draco_oxide_amd import gltf, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
glbs, total = synth.batch_glbs(n, seams=True)
gltf.transcode_files(glbs)
for _ in range(4):
    tm = {}
    t0 = time.perf_counter()
    gltf.transcode_files(glbs, timings=tm)
    dt = time.perf_counter() - t0
    print(f"{n} seam GLBs / {total} triangles: {dt * 1e3:.1f} ms = {total / dt / 1e6:.1f} Mtri/s; build {tm['build_s'] * 1e3:.1f}, prepare {tm['prepare_s'] * 1e3:.1f}, encode {tm['encode_s'] * 1e3:.1f}, assemble {tm['assemble_s'] * 1e3:.1f}", flush=True)
