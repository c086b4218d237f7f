#!/usr/bin/env python3
"""Two dmi_meshes_build calls side by side (threads), then prepare + encode of each: same blobs as one after the other?"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import draco_oxide_amd as dmi  # noqa: E402
from draco_oxide_amd import gltf, synth  # noqa: E402

glbs, total = synth.batch_glbs(int(sys.argv[1]) if len(sys.argv) > 1 else 256)
docs = [gltf.load_document(g) for g in glbs]
raws = []
for doc, bufs in docs:
    for prim, names, w in gltf._plan(doc):
        raws.append(gltf.primitive_to_raw(doc, bufs, prim)[0])
half = len(raws) // 2
parts = [raws[:half], raws[half:]]
cfg = dmi.Config.default()


def encode_part(raw_list):
    batch = dmi.meshes_build(raw_list, cfg)
    try:
        nf, npts = batch.counts()
        keep = [k for k in range(len(raw_list)) if nf[k] > 0]
        jobs = dmi.built_meshes_prepare(batch, keep, cfg)
    finally:
        batch.free()
    try:
        return [j.header_and_connectivity + s for j, s in zip(jobs, dmi.jobs_encode(jobs))]
    finally:
        for j in jobs:
            j.close()


ref = [encode_part(p) for p in parts]
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
    got, errs = [None, None], []

    def run(i):
        try:
            got[i] = encode_part(parts[i])
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    print("rep", rep, "errors:", [str(e)[:200] for e in errs], "same blobs:", got == ref, flush=True)
