#!/usr/bin/env python3
"""A/B of transcode settings inside ONE process (the GLBs are generated once, settings alternate call by call: box-to-box and run-to-run noise is
±7 %).  usage: transcode_ab.py VAR=value1,value2,... [reps]   e.g.  transcode_ab.py "DMI_STAGE_THREADS=16:16:16,2:12:2,2:14:2" 8   (':' stands for ',')"""
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import draco_oxide_amd as dmi  # noqa: E402
from draco_oxide_amd import gltf, synth  # noqa: E402

var, values = sys.argv[1].split("=", 1)
values = [v.replace(":", ",") for v in values.split(",")] if ":" in values else values.split(",")
if ":" in sys.argv[1]:
    values = [v.replace(":", ",") for v in sys.argv[1].split("=", 1)[1].split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
glbs, total = synth.batch_glbs(int(os.environ.get("AB_FILES", 1024)), seams=bool(os.environ.get("AB_SEAMS")))
gltf.transcode_files(glbs)
times = {v: [] for v in values}
for r in range(reps):
    for v in values:
        os.environ[var] = v
        t0 = time.perf_counter()
        gltf.transcode_files(glbs)
        times[v].append(time.perf_counter() - t0)
for v in values:
    t = times[v]
    print(f"{var}={v}: median {statistics.median(t) * 1e3:.1f} ms = {total / statistics.median(t) / 1e6:.1f} Mtri/s, best {min(t) * 1e3:.1f}, worst {max(t) * 1e3:.1f}", flush=True)
