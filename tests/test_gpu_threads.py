"""Calls from several host threads at once: whole-mesh encodes, batch prepares and transcodes side by side give the bytes they give alone (library
streams are pooled across threads, staging and device chunks are shared pools, the walkers of a batch start while its coordinator still packs)."""
import threading

import numpy as np
import pytest

import draco_oxide_amd as dmi
from draco_oxide_amd import gltf, synth

pytestmark = pytest.mark.gpu


def _run_threads(fns):
    out, errs = [None] * len(fns), []

    def run(i):
        try:
            out[i] = fns[i]()
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=run, args=(i,)) for i in range(len(fns))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]
    return out


def test_concurrent_calls_from_four_threads():
    big = [synth.torus_mesh(300 + 11 * k, seed=50 + k, open_boundary=bool(k & 1)) for k in range(2)]          # device tables (≥ 2^16 faces)
    dev = [dmi.DeviceMesh.upload(m, 0) for m in big]
    batch = synth.batch_meshes(24, lo=2e3, hi=4e4, seed=9)
    glbs, _ = synth.batch_glbs(12, lo=2e3, hi=3e4, seed=11)

    def whole(k):
        return lambda: [dmi.encode_mesh_device(dev[k]) for _ in range(3)]

    def prep():
        jobs = dmi.meshes_prepare(batch)
        try:
            return [j.header_and_connectivity + s for j, s in zip(jobs, dmi.jobs_encode(jobs))]
        finally:
            for j in jobs:
                j.close()

    def trans():
        return [bytes(g) for g, _ in gltf.transcode_files(glbs)]

    alone = [whole(0)(), whole(1)(), prep(), trans()]
    assert alone[0][0] == dmi.encode_mesh(big[0]) and alone[1][0] == dmi.encode_mesh(big[1])
    for _ in range(3):
        together = _run_threads([whole(0), whole(1), prep, trans])
        assert together == alone
    # threads that come and go: their library streams return to the pool and serve the next ones
    for _ in range(4):
        assert _run_threads([prep, trans]) == alone[2:]
