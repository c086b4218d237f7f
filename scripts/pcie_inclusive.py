"""PCIe-inclusive rate of the drop-in boundary: dmi_encode_attributes with host pointers in (job creation = relabelling,
uploads, fan rows; encode; read-back) on the bench workload, next to the resident form."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import draco_oxide_amd as dmi
from draco_oxide_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
mesh = synth.torus_mesh(n)
conn = dmi.encode_connectivity(mesh)
tables = [conn.table(i) for i in range(conn.num_tables)]
seeds = conn.seeds()
dmi.encode_attributes(mesh.attributes, tables, seeds=seeds)   # warm-up (HIP init, kernels)
t = []
for _ in range(3):
    t0 = time.perf_counter()
    out = dmi.encode_attributes(mesh.attributes, tables, seeds=seeds)
    t.append(time.perf_counter() - t0)
job = dmi.Job.from_tables(mesh.attributes, tables, seeds=seeds)
job.encode()
r = []
for _ in range(3):
    t0 = time.perf_counter()
    out2 = job.encode()
    r.append(time.perf_counter() - t0)
assert out == out2
print(json.dumps({"triangles": len(mesh.faces), "host_pointers_in_s": round(min(t), 4), "resident_s": round(min(r), 4),
                  "mtri_per_s_host_pointers_in": round(len(mesh.faces) / min(t) / 1e6, 2), "mtri_per_s_resident": round(len(mesh.faces) / min(r) / 1e6, 2)}))
