"""round 6: the exporter-style (UV seam) transcode at 256 / 512 / 1024 files — is the 256-file figure the bench prints a small-batch figure?
   python3 scripts/experiments/seams_scale.py [steps=5]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import draco_oxide_amd as dmi
from draco_oxide_amd import synth, gltf, binding

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cfg = dmi.Config(device=0)
if not os.environ.get("DMI_BENCH_PLAIN_PROCESS"):
    binding.configure_process(huge_page_new=True, numa_pin=True)
for n, seams in ((256, True), (512, True), (1024, True), (256, False), (1024, False)):
    glbs, total = synth.batch_glbs(n, seams=seams)
    alist = binding.AssetList(glbs)
    for _ in range(2):
        gltf.transcode_files(alist, cfg)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter(); gltf.transcode_files(alist, cfg); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(json.dumps({"files": n, "seams": seams, "triangles": int(total), "median_ms": round(ts[len(ts) // 2] * 1e3, 2), "min_ms": round(ts[0] * 1e3, 2),
                      "Mtri_per_s": round(total / ts[len(ts) // 2] / 1e6, 1)}), flush=True)
