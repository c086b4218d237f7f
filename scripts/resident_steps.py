"""A resident job of the 10M-triangle workload re-encoded a few times (the hot path alone: what the kernel profiles look at)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import draco_oxide_amd as dmi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
mesh = dmi.synth.torus_mesh(n)
job = dmi.mesh_prepare(mesh, dmi.Config(flags=dmi.FLAG_TIMINGS))
for _ in range(steps):
    job.encode_raw().free()
t = job.timings()
print({k: round(v, 4) if isinstance(v, float) else v for k, v in t.items() if k.endswith("_ms")})
job.close()
