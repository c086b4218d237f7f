"""round 6: do the library's waits (hipStreamSynchronize / hipEventSynchronize of its build / prepare / encode coordinators) burn CPU the walks could use?
hipSetDeviceFlags(hipDeviceScheduleBlockingSync) before anything runs, against the default.  python3 blocking_sync_ab.py [default|blocking|yield] [plain|seams] [calls=10]"""
import os, sys, time, resource, ctypes as C
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
kind = sys.argv[2] if len(sys.argv) > 2 else "plain"
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 10
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import draco_oxide_amd as dmi
from draco_oxide_amd import synth, gltf, binding
binding.load_library()
hip = C.CDLL("libamdhip64.so.7")
if mode != "default":
    rc = hip.hipSetDeviceFlags({"blocking": 4, "yield": 2, "spin": 1}[mode])
    print("hipSetDeviceFlags ->", rc)
binding.configure_process(huge_page_new=True, numa_pin=True)
glbs, total = synth.batch_glbs(1024, seams=(kind == "seams"))
alist = binding.AssetList(glbs)
cfg = dmi.Config(device=0)
for _ in range(3):
    gltf.transcode_files(alist, cfg)
ts, cpus = [], []
for _ in range(calls):
    r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter()
    gltf.transcode_files(alist, cfg)
    dt = time.perf_counter() - t0; r1 = resource.getrusage(resource.RUSAGE_SELF)
    ts.append(dt); cpus.append((r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime))
ts2 = sorted(ts)
print(f"{mode} {kind}: median {ts2[len(ts2) // 2] * 1e3:.1f} ms = {total / ts2[len(ts2) // 2] / 1e6:.0f} Mtri/s, min {ts2[0] * 1e3:.1f}; CPU per call median {sorted(cpus)[len(cpus) // 2] * 1e3:.0f} ms")
