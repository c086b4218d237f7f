"""round 6: the whole-mesh call of bench.py under the HIP runtime the torch wheel bundles (7.0: every host <-> device copy is a blit KERNEL that holds up the
other queues' kernels) and under the system's (7.2: SDMA engines).  python3 runtime_ab.py [torch|system] [grid=2236] [steps=9]"""
import os, sys, time, json
which = sys.argv[1] if len(sys.argv) > 1 else "torch"
if which == "system":
    os.environ["DMI_NO_TORCH_PREIMPORT"] = "1"
else:
    import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import draco_oxide_amd as dmi
from draco_oxide_amd import synth, binding
import ctypes as C
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 2236
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 9
binding.configure_process(huge_page_new=True, numa_pin=True)
mesh = synth.torus_mesh(grid)
dm = dmi.DeviceMesh.upload(mesh)
cfg = dmi.Config(flags=dmi.FLAG_TIMINGS)
for _ in range(3):
    ref = dmi.encode_mesh_device(dm, cfg)
ts, tabs = [], []
for _ in range(steps):
    t0 = time.perf_counter(); out = dmi.encode_mesh_device(dm, cfg); ts.append(time.perf_counter() - t0)
    assert out == ref
    tabs.append(dmi.last_call_timings())
ts2 = sorted(ts)
med = tabs[ts.index(ts2[len(ts2) // 2])]
v = C.c_int(); C.CDLL("libamdhip64.so.7").hipRuntimeGetVersion(C.byref(v))
print(json.dumps({"runtime": which, "hip_runtime_version": v.value, "torch_imported": "torch" in sys.modules, "median_ms": round(ts2[len(ts2) // 2] * 1e3, 2), "min_ms": round(ts2[0] * 1e3, 2),
                  "tables_ms": round(med["tables_ms"], 2), "connectivity_ms": round(med["connectivity_ms"], 2), "job_create_ms": round(med["job_create_ms"], 3), "encode_total_ms": round(med["total_ms"], 2),
                  "bytes": len(ref)}), flush=True)
