"""1024 GLBs through gltf.transcode_files, inputs in pageable memory and in dmi_host_alloc memory, alternating: python scripts/experiments/in_place_ab.py [rounds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import draco_oxide_amd as dmi
from draco_oxide_amd import binding, gltf, synth
glbs, total = synth.batch_glbs(1024)
held = [binding.HostBuffer.holding(g) for g in glbs]
views = [h.view() for h in held]
for _ in range(2):
    gltf.transcode_files(glbs); gltf.transcode_files(views)
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    for name, src in (("pageable", glbs), ("in place", views)):
        ts = []
        for _ in range(5):
            tm = {}
            t0 = time.perf_counter(); res = gltf.transcode_files(src, timings=tm); ts.append(time.perf_counter() - t0); del res
        st = tm.get("native", {})
        print(f"{name}: " + " ".join(f"{t * 1e3:.1f}" for t in ts) + f" ms; last call: in place {st.get('primitives_in_place')}, build {st.get('build_ms', 0):.1f} prepare {st.get('prepare_ms', 0):.1f} encode {st.get('encode_ms', 0):.1f} (summed)", flush=True)
