#!/usr/bin/env python3
"""Where does the run-to-run variance of the pipelined transcode come from?  16 calls in one process, split of each."""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import draco_oxide_amd as dmi  # noqa: E402
from draco_oxide_amd import gltf, synth  # noqa: E402

glbs, total = synth.batch_glbs(1024)
gltf.transcode_files(glbs)
if os.environ.get("NOGC"):
    gc.disable()
rows = []
for r in range(16):
    tm = {}
    t0 = time.perf_counter()
    res = gltf.transcode_files(glbs, timings=tm)
    dt = time.perf_counter() - t0
    del res
    rows.append((dt, tm))
    print(f"{dt * 1e3:7.1f} ms  parse {tm['parse_s'] * 1e3:5.1f} views {tm['views_s'] * 1e3:5.1f} build {tm['build_s'] * 1e3:6.1f} prepare {tm['prepare_s'] * 1e3:6.1f} encode {tm['encode_s'] * 1e3:6.1f} assemble {tm['assemble_s'] * 1e3:5.1f}", flush=True)
