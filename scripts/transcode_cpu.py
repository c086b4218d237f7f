#!/usr/bin/env python3
"""CPU time against wall time of gltf.transcode_files on the configs[3] batch (is the 16-CPU quota the limit?): python scripts/transcode_cpu.py [n_files=1024] [calls=5]"""
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import draco_oxide_amd as dmi  # noqa: E402
from draco_oxide_amd import gltf, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
glbs, total = synth.batch_glbs(n)
dmi.init(0)
for k in range(calls):
    tm = {"trace": []} if k == calls - 1 else {}
    r0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    out = gltf.transcode_files(glbs, timings=tm)
    dt = time.perf_counter() - t0
    r1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
    print(f"call {k}: {dt * 1e3:.1f} ms = {total / dt / 1e6:.1f} Mtri/s; CPU {cpu * 1e3:.0f} ms (user {(r1.ru_utime - r0.ru_utime) * 1e3:.0f}, sys {(r1.ru_stime - r0.ru_stime) * 1e3:.0f}) = {cpu / dt:.1f} CPUs busy; "
          f"involuntary switches {r1.ru_nivcsw - r0.ru_nivcsw}, minor faults {r1.ru_minflt - r0.ru_minflt}; " + ", ".join(f"{k2} {v * 1e3:.0f}" for k2, v in tm.items() if k2.endswith("_s") and k2 != "trace"), flush=True)
    del out
    if "trace" in tm:
        for step, first, a, b in sorted(tm["trace"], key=lambda x: x[2]):
            print(f"   {step:9s} stage@{first:5d}  {(a - t0) * 1e3:7.1f} -> {(b - t0) * 1e3:7.1f}  ({(b - a) * 1e3:5.1f} ms)")
try:
    print(open("/sys/fs/cgroup/cpu.stat").read())
except OSError:
    pass
