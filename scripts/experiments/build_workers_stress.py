#!/usr/bin/env python3
"""gltf.transcode_files with several build workers, call after call: how often does a call fail, with what?"""
import os
import sys
import time
import collections

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import draco_oxide_amd as dmi  # noqa: E402
from draco_oxide_amd import gltf, synth  # noqa: E402

glbs, total = synth.batch_glbs(1024)
ref = gltf.transcode_files(glbs)
errs = collections.Counter()
times = []
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    t0 = time.perf_counter()
    try:
        out = gltf.transcode_files(glbs)
        times.append(time.perf_counter() - t0)
        if [o[0] for o in out] != [o[0] for o in ref]:
            errs["different bytes"] += 1
    except Exception as e:  # noqa: BLE001
        errs[str(e)[:160]] += 1
times.sort()
print("calls ok:", len(times), "median ms: %.1f" % (times[len(times) // 2] * 1e3 if times else -1), "errors:", dict(errs))
