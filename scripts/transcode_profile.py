#!/usr/bin/env python3
"""The 1024-file transcode (bench.py transcode_regime's workload) N times, for a kernel trace: python scripts/transcode_profile.py [n_files] [calls] [seams]
(scripts/profile_transcode.sh runs it under rocprofv3 and writes profiles/<tag>_transcode_kernel_stats.csv: per-call kernel totals)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import draco_oxide_amd as dmi  # noqa: E402
from draco_oxide_amd import gltf, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 6
seams = len(sys.argv) > 3 and sys.argv[3] == "seams"   # exporter-style UV seams (bench.py transcode_regime.with_uv_seams: 256 files)
glbs, total = synth.batch_glbs(n, seams=seams)
ts = []
for k in range(calls):
    t0 = time.perf_counter()
    res = gltf.transcode_files(glbs)
    ts.append(time.perf_counter() - t0)
    del res
print(f"{n} files, {total} triangles, {calls} calls: " + " ".join(f"{t * 1e3:.1f}" for t in ts) + " ms")
