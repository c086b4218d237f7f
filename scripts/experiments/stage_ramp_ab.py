"""round 6: the transcoder's first stages — DMI_STAGE_RAMP (first stage = 1/n of a stage, doubling) against the default (a third, then whole stages), alternating in
one process: python3 scripts/experiments/stage_ramp_ab.py [plain|seams] [n_files=1024] [reps=5] [settings=0,6,12,24]"""
import os, sys, time, json, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import draco_oxide_amd as dmi
from draco_oxide_amd import synth, gltf, binding

kind = sys.argv[1] if len(sys.argv) > 1 else "plain"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
# settings: '/'-separated; each a ','-separated list of VAR=value (or a bare number = DMI_STAGE_RAMP=<n>, 0 = nothing set)
settings = [s for s in (sys.argv[4] if len(sys.argv) > 4 else "0/6/12/24").split("/")]
var = "setting"
def apply(s):
    for k in ("DMI_STAGE_RAMP", "DMI_SMALL_HEAD", "DMI_STAGE_PRIMITIVES", "DMI_FILE_ORDER", "DMI_SPIN_WAITS", "DMI_NO_QUAD", "DMI_NO_STREAM_COPY", "DMI_PREPARE_THREADS"):
        os.environ.pop(k, None)
    if s in ("0", ""):
        return
    for kv in s.split(","):
        k, _, v = kv.partition("=")
        if not v:
            k, v = "DMI_STAGE_RAMP", k
        os.environ[k] = v
binding.configure_process(huge_page_new=True, numa_pin=True)
glbs, total = synth.batch_glbs(n, seams=(kind == "seams"))
alist = binding.AssetList(glbs)
for _ in range(2):
    gltf.transcode_files(alist, dmi.Config(device=0))
res = {s: [] for s in settings}
cpu = {s: [] for s in settings}
for rnd in range(3):
    for s in settings:
        apply(s)
        cfg = dmi.Config(device=0)   # (the binding fills dmi_debug from the environment per Config)
        gltf.transcode_files(alist, cfg)
        for _ in range(reps):
            r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.perf_counter(); gltf.transcode_files(alist, cfg); res[s].append(time.perf_counter() - t0)
            r1 = resource.getrusage(resource.RUSAGE_SELF); cpu[s].append((r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime))
for s in settings:
    ts = sorted(res[s])
    print(json.dumps({"kind": kind, "files": n, var: s, "median_ms": round(ts[len(ts) // 2] * 1e3, 2), "min_ms": round(ts[0] * 1e3, 2), "q1_ms": round(ts[len(ts) // 4] * 1e3, 2),
                      "Mtri_per_s": round(total / ts[len(ts) // 2] / 1e6, 1), "cpu_ms_median": round(sorted(cpu[s])[len(cpu[s]) // 2] * 1e3)}), flush=True)
