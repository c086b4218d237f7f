"""End-to-end batch transcode rate with the host connectivity stage of batch k+1 overlapped with the device encode of batch k.
dmi_meshes_prepare (host graph walks + uploads, a pool of host threads) and dmi_jobs_encode (device) are independent, thread-safe
C-ABI calls: a caller pipelines them from two threads — no extra entry point is needed.  BASELINE configs[3] shape: 1024 meshes, F
log-uniform in [2k, 200k], pos+nrm+uv; mesh in → whole .drc out for every mesh.
usage: python scripts/bench_pipeline.py [meshes=1024] [batches=4]"""
import json, os, sys, threading, time, queue
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import draco_oxide_amd as dmi
from draco_oxide_amd import synth

n_meshes = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 4
meshes = synth.batch_meshes(n_meshes)
total = sum(len(m.faces) for m in meshes)
parts = [meshes[k::n_batches] for k in range(n_batches)]           # similar triangle counts per part
warm = dmi.meshes_prepare(parts[0][:8]); dmi.jobs_encode(warm); [j.close() for j in warm]   # HIP init, kernels, pools


def serial():
    t0 = time.perf_counter()
    nbytes = 0
    for p in parts:
        jobs = dmi.meshes_prepare(p)
        with dmi.jobs_encode_raw(jobs) as out:
            nbytes += out.nbytes + sum(len(j.header_and_connectivity) for j in jobs)
        for j in jobs:
            j.close()
    return time.perf_counter() - t0, nbytes


def pipelined():
    q = queue.Queue(maxsize=2)
    def producer():
        for p in parts:
            q.put(dmi.meshes_prepare(p))
        q.put(None)
    t0 = time.perf_counter()
    th = threading.Thread(target=producer)
    th.start()
    nbytes = 0
    while True:
        jobs = q.get()
        if jobs is None:
            break
        with dmi.jobs_encode_raw(jobs) as out:
            nbytes += out.nbytes + sum(len(j.header_and_connectivity) for j in jobs)
        for j in jobs:
            j.close()
    th.join()
    return time.perf_counter() - t0, nbytes


s_t, s_b = min(serial() for _ in range(2))
p_t, p_b = min(pipelined() for _ in range(2))
assert s_b == p_b
one = time.perf_counter(); jobs = dmi.meshes_prepare(meshes); prep = time.perf_counter() - one
one = time.perf_counter(); dmi.jobs_encode_raw(jobs).free(); enc = time.perf_counter() - one
for j in jobs:
    j.close()
print(json.dumps({"workload": f"{n_meshes} meshes, F log-uniform [2k,200k], pos+nrm+uv, mesh in -> .drc out", "triangles": int(total), "drc_bytes": int(s_b),
                  "one_batch": {"prepare_s": round(prep, 3), "encode_s": round(enc, 4), "mtri_per_s": round(total / (prep + enc) / 1e6, 1)},
                  "serial_batches": {"batches": n_batches, "seconds": round(s_t, 3), "mtri_per_s": round(total / s_t / 1e6, 1)},
                  "pipelined_batches": {"batches": n_batches, "seconds": round(p_t, 3), "mtri_per_s": round(total / p_t / 1e6, 1)}}))
