"""round 6: blocking_sync_ab.py with the two settings alternating inside ONE process (boxes and runs differ by more than the effect).
python3 blocking_sync_ab2.py [plain|seams] [reps=6] [rounds=4]"""
import os, sys, time, resource, ctypes as C
kind = sys.argv[1] if len(sys.argv) > 1 else "plain"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import draco_oxide_amd as dmi
from draco_oxide_amd import synth, gltf, binding
binding.load_library()
hip = C.CDLL("libamdhip64.so.7")
binding.configure_process(huge_page_new=True, numa_pin=True)
glbs, total = synth.batch_glbs(1024, seams=(kind == "seams"))
alist = binding.AssetList(glbs)
cfg = dmi.Config(device=0)
for _ in range(3):
    gltf.transcode_files(alist, cfg)
res = {"auto": ([], []), "blocking": ([], [])}
for rnd in range(rounds):
    for name, flag in (("auto", 0), ("blocking", 4)):
        assert hip.hipSetDeviceFlags(flag) == 0
        gltf.transcode_files(alist, cfg)
        for _ in range(reps):
            r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter()
            gltf.transcode_files(alist, cfg)
            dt = time.perf_counter() - t0; r1 = resource.getrusage(resource.RUSAGE_SELF)
            res[name][0].append(dt); res[name][1].append((r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime))
for name, (ts, cpus) in res.items():
    ts2 = sorted(ts)
    print(f"{kind} {name}: median {ts2[len(ts2) // 2] * 1e3:.1f} ms = {total / ts2[len(ts2) // 2] / 1e6:.0f} Mtri/s, q1 {ts2[len(ts2) // 4] * 1e3:.1f}, min {ts2[0] * 1e3:.1f}; CPU per call median {sorted(cpus)[len(cpus) // 2] * 1e3:.0f} ms", flush=True)
