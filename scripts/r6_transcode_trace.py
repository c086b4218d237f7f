#!/usr/bin/env python3
"""Stage timeline + kernel totals of the 1024-file transcode (plain and exporter-style seams): python scripts/r6_transcode_trace.py [plain|seams] [n]
Run under DMI_TRACE_STAGES=1 for the library's stage lines on stderr."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import draco_oxide_amd as dmi  # noqa: E402
from draco_oxide_amd import gltf, synth  # noqa: E402

dmi.configure_process(huge_page_new=True, numa_pin=True)   # (as bench.py does)
kind = sys.argv[1] if len(sys.argv) > 1 else "plain"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (1024 if kind == "plain" else 256)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
glbs, total = synth.batch_glbs(n, seams=(kind == "seams"))
trace = os.environ.pop("DMI_TRACE_STAGES", None)
for _ in range(2):
    gltf.transcode_files(glbs)
ts = []
for r in range(reps):
    if trace and r == reps - 1:
        os.environ["DMI_TRACE_STAGES"] = "1"
        print(f"---- traced call ({kind}, {n} files, {total} triangles)", file=sys.stderr, flush=True)
    tm = {}
    t0 = time.perf_counter()
    gltf.transcode_files(glbs, timings=tm)
    dt = time.perf_counter() - t0
    ts.append(dt)
    st = tm.get("native", {})
    print(f"{kind} {n} files / {total} tri: {dt * 1e3:.1f} ms = {total / dt / 1e6:.1f} Mtri/s; parse {st.get('parse_ms', 0):.1f} pushed {st.get('pushed_ms', 0):.1f} finished {st.get('finished_ms', 0):.1f} "
          f"build {tm['build_s'] * 1e3:.1f} prepare {tm['prepare_s'] * 1e3:.1f} encode {tm['encode_s'] * 1e3:.1f} assemble {tm['assemble_s'] * 1e3:.1f}", flush=True)
ts.sort()
print(f"{kind}: median {ts[len(ts) // 2] * 1e3:.1f} ms = {total / ts[len(ts) // 2] / 1e6:.1f} Mtri/s, min {ts[0] * 1e3:.1f}", flush=True)
