import os, sys, time
os.environ.setdefault("DMI_TRACE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import draco_oxide_amd as d
from importlib import import_module
synth = import_module("draco-oxide_amd.synth") if False else d.synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
m = synth.torus_mesh(n)
print(type(m), len(m.faces))
for r in range(5):
    t0 = time.perf_counter()
    c = d.encode_connectivity(m)
    print("encode_connectivity %.1f ms" % ((time.perf_counter() - t0) * 1e3))
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "|", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
for line in open("/proc/self/smaps_rollup"):
    if line.startswith(("Rss", "AnonHuge")):
        print(line.strip())
