import os, sys, time
sys.path.insert(0, os.getcwd())
import draco_oxide_amd as dmi
from draco_oxide_amd import gltf, synth
glbs, total = synth.batch_glbs(256, seams=True)
ts = []
for k in range(6):
    t0 = time.perf_counter(); res = gltf.transcode_files(glbs); ts.append(time.perf_counter() - t0); del res
print(f"256 seam files, {total} triangles: " + " ".join(f"{t*1e3:.1f}" for t in ts) + " ms", file=sys.stderr)
