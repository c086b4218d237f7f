#!/usr/bin/env python3
"""bench.py — draco-oxide's encode path on MI355X: mesh in HBM → whole `.drc` on the host.

One "step" = one `dmi_encode_mesh_device` call (= the reference's `encode::encode(mesh, &mut buf, Config::default())`, encode/mod.rs:59)
on one mesh whose faces and attribute values are resident in HBM when the timed region starts: device corner tables (half-edge
matching, left-most corners, boundary flags) → read-back for the two serial host walks (Edgebreaker traversal, attribute sequencer) →
coding-order relabelling + fan rows on the device → the attribute-encoding hot path (value ranges → coding-order gather + quantize →
predict + transform → histograms → table stage → rANS/rABS stream coding) → spliced `.drc` bytes in a library-owned host buffer.
At N > 1 every rank encodes its own mesh (independent meshes shard with no data-path collective) and the finished bitstreams are
gathered onto rank 0 over RCCL inside the timed region (weak scaling); rank 0 additionally reports `batch_sharded` = BASELINE configs[3]
dealt over the N ranks (strong scaling, prepare inside the timed region).

Workload (BASELINE.json configs[2], the configuration the metric's target is quoted on): 10M-triangle synthetic closed torus grid
(n=2236 → 9 999 392 triangles, 4 999 696 vertices), positions + normals + UVs, Edgebreaker order, parallelogram / normal /
texcoord prediction, wrapped-difference + octahedral transforms, 11/8/10-bit quantization — `encode::Config::default()`.

Other scopes on the line: `resident_attribute_step` = `dmi_job_encode` on a resident job (the hot path alone, what rounds 1–2 reported as
`value`); `boundary_call` = `dmi_encode_attributes` with host pointers in (what the Rust shim binds); `end_to_end_host_memory` =
`dmi_encode_mesh` from host memory (PCIe-inclusive); `batch_regime` = 256 meshes through dmi_meshes_prepare + dmi_jobs_encode.
`roofline` = the quantize+predict pass of the hot path, hipEvent-timed on the stream it is launched on inside the timed steps.
`cpu_baseline` = the oracle (CPU restatement of the reference, one core) on the same mesh with its per-stage split.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import draco_oxide_amd as dmi  # noqa: E402
from draco_oxide_amd import distributed as dmi_dist  # noqa: E402
from draco_oxide_amd import synth  # noqa: E402

HBM_PEAK_GBPS = 8000.0         # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s)
HBM_ACHIEVABLE_GBPS = 6290.0   # the same guide's float4 copy kernel: what a pure streaming kernel reaches on this part


def usable_cpus():
    """CPUs this process may actually use: its affinity mask, capped by the cgroup's CPU quota (a container on a 256-thread host may be
    allowed 16 CPUs' worth of time)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


PASS_KERNEL_PREFIXES = ("k_value_ranges", "k_value_quantize_rec", "k_seq_quantize", "k_i32_minmax_final", "k_predict_fused", "k_predict_packed", "k_orient_summary",
                        "k_pred_parallelogram", "k_pred_texcoord", "k_pred_delta", "k_seq_gather_rec", "k_texcoord_fixup")


def measure_traffic(grid):
    """HBM bytes of the quantize+predict pass per step from the PMC counters, measured by THIS run: two child processes — `rocprofv3 --pmc FETCH_SIZE` and
    `--pmc WRITE_SIZE` (the two cannot share a pass: MI355X_MICROARCH.md) around `python3 bench.py --steps 2 --warmup 1` without the side measurements — started
    BEFORE this process touches the GPU.  Counter unit KiB; FETCH_SIZE doubled (gfx950 reports half of a wide coalesced read: exact for the streaming kernels,
    an upper bound for the gather kernels); summed over every kernel that quantizes or predicts (early stage included), per step.  None when rocprofv3 is not
    there or fails (the line then carries `traffic_profiled`: the last committed session's figure)."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None
    steps, warm = 2, 1
    sums = {}
    tmp = tempfile.mkdtemp(prefix="dmi_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [rp, "--pmc", counter, "-d", out, "-o", "p", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warm),
                   "--grid", str(grid), "--no-cpu-baseline", "--no-batch", "--no-scopes", "--no-traffic"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
            hits = glob.glob(out + "/**/*_counter_collection.csv", recursive=True)
            if r.returncode != 0 or not hits:
                return None
            tot = 0.0
            for row in csv.DictReader(open(hits[0])):
                m = re.search(r"(k_[a-z_0-9]+)", row["Kernel_Name"])
                if row["Counter_Name"] == counter and m and m.group(1).startswith(PASS_KERNEL_PREFIXES):
                    tot += float(row["Counter_Value"])
            sums[counter] = tot * 1024 / (steps + warm)
        return int(2 * sums["FETCH_SIZE"] + sums["WRITE_SIZE"])
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def profiled_traffic():
    """The pass's HBM traffic as the last committed profiling session measured it (profiles/round*_bench.json: FETCH_SIZE / WRITE_SIZE in separate
    rocprofv3 --pmc passes, scripts/profile_round.sh) — NOT measured in this run (`traffic` is, when the run is part of such a session)."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "round*_bench.json"))):
        try:
            t = json.load(open(path)).get("roofline", {}).get("traffic")
        except (OSError, ValueError):
            continue
        if t:
            best = {"bytes": int(t), "source": "profiles/" + os.path.basename(path), "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of that session (profiles/round*_pmc_traffic.csv)"}
    return best


def cpu_baseline(mesh):
    """Reference algorithm on one host core: the oracle (CPU restatement, kind "port"), timed on the SAME mesh and the SAME scope as
    `value` (whole `.drc`), with the per-stage split of BASELINE.md §3.  The reference's own complexity (`faithful`: linear `contains`
    scans, O(V²)) is timed on a bounded ≈100k-triangle sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers  # test infrastructure: allowed in the cpu_baseline leg only
    sess = helpers.oracle_from_product_mesh(mesh)
    t0 = time.time()
    sess.encode(dump=False)
    wall = time.time() - t0
    split = sess.stage_split()
    scope_s = max(split["attribute_section_s"] - split["sequencer_s"], 1e-9)
    f = len(mesh.faces)
    out = {
        "value": round(f / wall / 1e6, 4), "unit": "Mtriangles/s", "cores": 1, "kind": "port", "host_cpus_usable": usable_cpus(),
        "sample": f"whole workload mesh ({f} triangles), oracle `ranked` mode (same bytes, O(1) already-coded test), whole .drc = {wall:.2f} s "
                  f"(the scope of `value`); attribute section minus sequencer alone = {scope_s:.2f} s = {f / scope_s / 1e6:.3f} Mtri/s (the scope of `resident_attribute_step`)",
        "attribute_section_mtri_per_s": round(f / scope_s / 1e6, 4),
        "stages_s": {k: round(v, 4) for k, v in split.items() if k.endswith("_s")},
        "rans_only_msym_per_s": round(split["rans_only_msym_per_s"], 2),
    }
    try:   # the reference's own complexity on a bounded sample (n=224 → 100 352 triangles, pos+nrm+uv)
        small = synth.torus_mesh(224)
        s2 = helpers.oracle_from_product_mesh(small)
        t0 = time.time()
        a = s2.encode(faithful=True, dump=False)
        t_f = time.time() - t0
        t0 = time.time()
        b = s2.encode(dump=False)
        t_r = time.time() - t0
        fs = len(small.faces)
        out["faithful_sample"] = {"triangles": fs, "faithful_s": round(t_f, 3), "ranked_s": round(t_r, 3), "same_bytes": a == b,
                                  "faithful_mtri_per_s": round(fs / t_f / 1e6, 4),
                                  "note": "faithful = the reference's linear `contains` scans and Vec::remove stack deletions: O(V^2); "
                                          f"extrapolated to the {f}-triangle workload ≈ {t_r + (t_f - t_r) * (f / fs) ** 2:.0f} s"}
    except Exception as e:   # never fails the line
        out["faithful_sample"] = {"error": str(e)[:200]}
    return out


def _med_min(ts):
    ts = sorted(ts)
    return ts[len(ts) // 2], ts[0]


def batch_regime(n_meshes=256, steps=5, device=0):
    """The batch form of the same path (BASELINE configs[3] shape): n independent meshes, F log-uniform in [2k, 200k], pos+nrm+uv.
    `value` = end to end: dmi_meshes_prepare (device tables for all meshes in one launch per kernel, host walks on the library's
    threads, batched relabelling) + ONE dmi_jobs_encode + dmi_free_many, per step, host meshes in → `.drc` pieces out.
    `resident_*` = the encode call alone on the resident jobs (what rounds 1–2 reported)."""
    meshes = synth.batch_meshes(n_meshes)
    total = sum(len(m.faces) for m in meshes)
    cfg = dmi.Config(device=device)
    jobs = dmi.meshes_prepare(meshes, cfg)   # warm-up: pools, staging, streams
    with dmi.jobs_encode_raw(jobs):
        pass
    for j in jobs:
        j.close()
    prep, enc = [], []
    nbytes = 0
    for _ in range(steps):
        t0 = time.perf_counter()
        jobs = dmi.meshes_prepare(meshes, cfg)
        t1 = time.perf_counter()
        with dmi.jobs_encode_raw(jobs) as batch:
            nbytes = batch.nbytes + sum(len(j.header_and_connectivity) for j in jobs)
        t2 = time.perf_counter()
        prep.append(t1 - t0)
        enc.append(t2 - t1)
        if _ + 1 < steps:
            for j in jobs:
                j.close()
    # the resident re-encode of the last step's jobs
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        with dmi.jobs_encode_raw(jobs) as batch:
            pass
    dt = (time.perf_counter() - t0) / steps
    with dmi.jobs_encode_raw(jobs) as batch:
        assert batch[0] == jobs[0].encode() and batch[n_meshes - 1] == jobs[n_meshes - 1].encode()
        ok = jobs[0].header_and_connectivity + batch[0] == dmi.encode_mesh(meshes[0], cfg)
    for j in jobs:
        j.close()
    e2e, e2e_min = _med_min([p + e for p, e in zip(prep, enc)])
    return {"workload": f"{n_meshes} independent meshes, F log-uniform [2k,200k], pos+nrm+uv: dmi_meshes_prepare + one dmi_jobs_encode per step, host meshes in, .drc pieces out",
            "triangles": int(total), "value": round(total / e2e / 1e6, 2), "unit": "Mtriangles/s", "statistic": f"median of {steps} steps", "ms_per_batch": round(e2e * 1e3, 3), "ms_per_batch_min": round(e2e_min * 1e3, 3),
            "prepare_ms": round(_med_min(prep)[0] * 1e3, 3), "encode_ms_after_prepare": round(_med_min(enc)[0] * 1e3, 3),
            "resident_ms_per_batch": round(dt * 1e3, 3), "resident_mtri_per_s": round(total / dt / 1e6, 2), "bitstream_bytes": int(nbytes),
            "sample_equals_single_mesh_encode": bool(ok)}


def transcode_regime(n_files=1024, steps=5, device=0):
    """BASELINE configs[3] as it is worded — "batch of 1024 glTF/glb meshes through KHR_draco_mesh_compression transcode": n GLB files
    (in memory, the reference's transcode_buffer form; F log-uniform in [2k, 200k], pos+nrm+uv, u16 / u32 indices) →
    gltf.transcode_files → n Draco-compressed GLBs.  Inside the timed call: container + JSON parse, primitive plans, accessor descriptors,
    MeshBuilder::build on the device for every primitive, connectivity stage + job creation, dmi_jobs_encode, the files written — all inside
    the library (dmi_transcode_assets, csrc/dmi_gltf.cpp; the interpreter only hands the list over).  `value` is the MEDIAN of `steps` calls
    (the minimum beside it) with the inputs in ordinary pageable memory (packed into staging by host threads, one copy up per group);
    `inputs_read_into_dmi_host_alloc_memory` is the same list held in the library's page-locked blocks (an importer that reads its files into
    dmi_host_alloc memory: accessors go up by DMA where they lie)."""
    from draco_oxide_amd import binding, gltf
    glbs, total = synth.batch_glbs(n_files)
    in_bytes = sum(len(g) for g in glbs)
    cfg = dmi.Config(device=device)
    for _ in range(2):
        gltf.transcode_files(glbs, cfg)                 # warm-up: staging, device pools, streams, the library's arenas
    # the list handed over as the C ABI takes it (binding.AssetList: the array of dmi_gltf_asset, made once — what DeviceMesh is to a mesh): the timed call is
    # dmi_transcode_assets + the views of its result; `python_list_form` below is the same call marshalling a Python list of bytes objects every time
    alist = binding.AssetList(glbs)
    ts, tms, res = [], [], None
    for _ in range(steps):
        tm = {}
        t0 = time.perf_counter()
        res = gltf.transcode_files(alist, cfg, timings=tm)
        ts.append(time.perf_counter() - t0)
        tms.append(tm)
    med, best = _med_min(ts)
    tpy = []
    for _ in range(3):
        t0 = time.perf_counter()
        gltf.transcode_files(glbs, cfg)
        tpy.append(time.perf_counter() - t0)
    tm = tms[ts.index(sorted(ts)[len(ts) // 2])]
    res1 = gltf.transcode_files(glbs[: n_files // 8 + 1], cfg, pipeline=False)   # one stage, the interpreter's loop: the same files
    # a sample of the embedded blobs against whole-mesh encodes of the host-built meshes (dmi_mesh_build), which the tests hold against the oracle
    ok = True
    for i in sorted({0, n_files // 3, n_files // 2, n_files - 1}):
        doc, binary = gltf.read_glb(glbs[i])
        mesh, _ = gltf.primitive_to_mesh(doc, binary, doc["meshes"][0]["primitives"][0])
        ok = ok and res[i][1][0] == dmi.encode_mesh(mesh, cfg)
    ok = ok and all(a[0] == b[0] for a, b in zip(res1, res))
    out_bytes = sum(len(g) for g, _ in res)
    st = tm.get("native", {})
    del res, res1
    # the same files read into dmi_host_alloc memory beforehand (an importer that reads its files into the library's page-locked blocks): accessors go
    # up where they lie — no host pack, no staging copy
    locked = None
    try:
        held = [binding.HostBuffer.holding(g) for g in glbs]
        views = [h.view() for h in held]
        gltf.transcode_files(views, cfg)
        tl, tml = [], []
        for _ in range(steps):
            tm2 = {}
            t0 = time.perf_counter()
            gltf.transcode_files(views, cfg, timings=tm2)
            tl.append(time.perf_counter() - t0)
            tml.append(tm2.get("native", {}))
        del views
        for h in held:
            h.free()
        m2, b2 = _med_min(tl)
        s2 = tml[tl.index(sorted(tl)[len(tl) // 2])]
        locked = {"value": round(total / m2 / 1e6, 2), "unit": "Mtriangles/s", "ms_per_batch_median": round(m2 * 1e3, 2), "ms_per_batch_min": round(b2 * 1e3, 2),
                  "buffers_in_place": int(s2.get("buffers_in_place", 0)), "pushed_ms": round(s2.get("pushed_ms", 0), 2), "finished_ms": round(s2.get("finished_ms", 0), 2)}
    except Exception as e:
        locked = {"error": str(e)[:200]}
    seam = None
    try:   # the same shape of files the way exporters write them: repeated positions / normals along the closing curves, a UV seam there.  Round 6: the SAME
        # list length as the plain figure (n_files; rounds 4–5 timed a quarter of it, and a call's ≈ 20 ms of pipeline fill and drain weigh four times as much
        # on a quarter of the triangles: 256 plain files run at 320 Mtri/s where 1024 run at 610) — the quarter-size figure stays beside it as `quarter_list`
        def seam_leg(n_s):
            sglbs, stotal = synth.batch_glbs(n_s, seams=True)
            sl = binding.AssetList(sglbs)
            for _ in range(2):
                gltf.transcode_files(sl, cfg)
            tsm = []
            for _ in range(steps):
                t0 = time.perf_counter()
                sres = gltf.transcode_files(sl, cfg)
                tsm.append(time.perf_counter() - t0)
            doc, binary = gltf.read_glb(sglbs[n_s // 2])
            mesh, _ = gltf.primitive_to_mesh(doc, binary, doc["meshes"][0]["primitives"][0])
            ms_, bs_ = _med_min(tsm)
            same = bool(bytes(sres[n_s // 2][1][0]) == dmi.encode_mesh(mesh, cfg))
            del sres
            return {"files": n_s, "triangles": int(stotal), "value": round(stotal / ms_ / 1e6, 2), "unit": "Mtriangles/s", "ms_per_batch_median": round(ms_ * 1e3, 2),
                    "ms_per_batch_min": round(bs_ * 1e3, 2), "sample_blob_equals_whole_mesh_encode": same}
        seam = seam_leg(n_files)
        seam["what"] = ("positions / normals repeated along the closing curves (merged by the device MeshBuilder), texture coordinates with a seam there: the UV attribute has a "
                        "corner table of its own; the same number of files as the plain figure")
        if n_files >= 32:
            seam["quarter_list"] = seam_leg(max(8, n_files // 4))
    except Exception as e:
        seam = {"error": str(e)[:200]}
    return {"with_uv_seams": seam, "inputs_read_into_dmi_host_alloc_memory": locked,
            "workload": f"BASELINE configs[3]: {n_files} GLB files in memory (one primitive each, F log-uniform [2k,200k], pos+nrm+uv, u16/u32 indices) → gltf.transcode_files → {n_files} "
                        "Draco-compressed GLBs: container + JSON parse, accessor descriptors, device MeshBuilder::build, dmi_built_meshes_prepare, dmi_jobs_encode, file assembly — "
                        "all inside dmi_transcode_assets, inside the timed call; inputs in pageable memory; the list handed over as a binding.AssetList "
                        "(the C array of dmi_gltf_asset made once, outside the timed calls — `python_list_form` marshals a list of bytes objects per call)",
            "triangles": int(total), "value": round(total / med / 1e6, 2), "unit": "Mtriangles/s", "statistic": f"median of {steps} calls", "ms_per_batch": round(med * 1e3, 2), "ms_per_batch_min": round(best * 1e3, 2),
            "python_list_form": {"ms_per_batch_median": round(_med_min(tpy)[0] * 1e3, 2), "value": round(total / _med_min(tpy)[0] / 1e6, 2)},
            "split_ms": {"parse (containers, JSON, plans, accessor descriptors; %d pool threads; wall clock to the last file parsed)" % int(st.get("parse_threads", 1)): round(st.get("parse_ms", 0), 2),
                         "parse, summed over its threads": round(st.get("parse_cpu_ms", 0), 2), "stages": int(st.get("stages", 0)),
                         "last primitive pushed at": round(st.get("pushed_ms", 0), 2), "last stage coded at": round(st.get("finished_ms", 0), 2),
                         "build (ingest / pack, kernels, faces + maps back; summed over stages, two threads)": round(st.get("build_ms", 0), 2),
                         "prepare (device tables, host walks, relabelling; summed, two threads)": round(st.get("prepare_ms", 0), 2),
                         "encode (summed)": round(st.get("encode_ms", 0), 2),
                         "assemble (library threads, summed)": round(st.get("assemble_ms", 0), 2), "library call": round(st.get("call_ms", 0), 2)},
            "stage_loop": "inside the library (dmi_transcode_assets → one dmi_transcoder per device: two build threads, two prepare threads sharing one budget of running walks, an encode thread; files written by library threads)",
            "input_bytes": int(in_bytes), "output_bytes": int(out_bytes),
            "primitives": {"built_by_the_device_kernels": int(st.get("primitives_device_built", 0)), "built_by_the_host_builder_inside_the_call": int(st.get("primitives_host_built", 0)),
                           "copied_up_in_place": int(st.get("primitives_in_place", 0))},
            "sample_blobs_equal_whole_mesh_encodes_and_one_stage_files": bool(ok)}


def batch_sharded(n_meshes, rank, world, local_rank, gather_dev, steps=2):
    """BASELINE configs[3] over the N ranks of this run: the batch dealt by triangle count (LPT); per step every rank PREPARES its share
    (device tables + host walks + relabelling: dmi_meshes_prepare) and encodes it (one dmi_jobs_encode), and the finished `.drc` blobs are
    gathered onto rank 0 in mesh order (RCCL).  Strong scaling, host meshes in → blobs on rank 0; rank 0 checks a sample of the gathered
    blobs byte for byte against single-mesh encodes."""
    meshes = synth.batch_meshes(n_meshes)
    weights = [len(m.faces) for m in meshes]
    mine = dmi_dist.shard_indices(n_meshes, rank, world, weights=weights)
    my_meshes = [meshes[i] for i in mine]
    cfg = dmi.Config(device=local_rank)
    times = {"prepare": 0.0, "encode": 0.0}

    def step(record=False):
        blobs = []
        if my_meshes:
            t0 = time.perf_counter()
            jobs = dmi.meshes_prepare(my_meshes, cfg)
            t1 = time.perf_counter()
            try:
                with dmi.jobs_encode_raw(jobs) as out:
                    blobs = [j.header_and_connectivity + out[k] for k, j in enumerate(jobs)]
            finally:
                for j in jobs:
                    j.close()
            if record:
                times["prepare"] += t1 - t0
                times["encode"] += time.perf_counter() - t1
        return dmi_dist.gather_blob_lists(blobs, mine, n_meshes, device=gather_dev)

    step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        got = step(record=True)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, times["prepare"] / steps, times["encode"] / steps], dtype=torch.float64, device=gather_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, prepare_max, encode_max = (float(x) for x in t.tolist())
    res = None
    if rank == 0:
        total = sum(weights)
        check = sorted(set([int(np.argmax(weights)), 0, n_meshes - 1] + list(range(0, n_meshes, max(1, n_meshes // 16)))))
        ok = True
        for i in check:
            ok = ok and bytes(got[i]) == dmi.encode_mesh(meshes[i], cfg)
        res = {"workload": f"BASELINE configs[3]: {n_meshes} meshes, F log-uniform [2k,200k], pos+nrm+uv, dealt over {world} rank(s) by triangle count (LPT); per step and rank: "
                           "dmi_meshes_prepare + one dmi_jobs_encode, blobs gathered on rank 0 (prepare INSIDE the timed region)", "scaling": "strong", "n_gpus": world,
               "triangles": int(total), "ms_per_step": round(dt / steps * 1e3, 3), "value": round(total * steps / dt / 1e6, 2), "unit": "Mtriangles/s",
               "prepare_ms_max_over_ranks": round(prepare_max * 1e3, 3), "encode_ms_max_over_ranks": round(encode_max * 1e3, 3),
               "host_threads_per_rank": int(os.environ.get("DMI_HOST_THREADS", usable_cpus())),
               "blobs_on_rank0": len(got), "sample_checked_against_single_encodes": len(check), "sample_ok": bool(ok)}
    return res


def transcode_sharded(n_files, rank, world, local_rank, gather_dev, steps=2):
    """BASELINE configs[3] over the N ranks of this run, as it is worded: n GLB files through gltf.transcode_files in the torch.distributed job (round 6) —
    the FILES are dealt to the ranks by their size in bytes (LPT) before anything is parsed, each rank runs dmi_transcode_assets over its own files (parse
    pool, build / prepare / encode pipeline, assembly), the finished files are gathered onto rank 0 (RCCL): no JSON is parsed on a rank that does not own
    it, nothing is reassembled on rank 0.  Strong scaling."""
    from draco_oxide_amd import gltf
    glbs, total = synth.batch_glbs(n_files)
    cfg = dmi.Config(device=local_rank)
    gltf.transcode_files(glbs, cfg, device=gather_dev)
    dist.barrier()
    torch.cuda.synchronize()
    owned, gather_s, own_s = 0, 0.0, 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        tm = {}
        res = gltf.transcode_files(glbs, cfg, device=gather_dev, timings=tm)
        owned = tm.get("files_owned", 0)
        gather_s += tm.get("gather_s", 0.0)
        own_s += tm.get("transcode_s", 0.0)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, float(owned), -float(owned), gather_s / steps, own_s / steps], dtype=torch.float64, device=gather_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, own_max, own_min, g_max, o_max = float(t[0]), int(t[1]), int(-t[2]), float(t[3]), float(t[4])
    # the same job with the finished files left on the ranks that made them (gather="manifest": a transcoder's outputs are files — each rank writes its own —, sizes and
    # digests of all files reach every rank in one all_reduce): what an N-GPU job costs when nothing but a manifest has to cross the ranks
    gltf.transcode_files(glbs, cfg, device=gather_dev, gather="manifest")
    dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    man = None
    for _ in range(steps):
        tm2 = {}
        kept = gltf.transcode_files(glbs, cfg, device=gather_dev, timings=tm2, gather="manifest")
        man = tm2.get("manifest")
    torch.cuda.synchronize()
    dist.barrier()
    dt_m = time.perf_counter() - t1
    tmm = torch.tensor([dt_m], dtype=torch.float64, device=gather_dev)
    dist.all_reduce(tmm, op=dist.ReduceOp.MAX)
    dt_m = float(tmm[0])
    if rank != 0:
        return None
    import xxhash
    mine_ok = all(e is None or (len(e[0]) == int(man[0][i]) and xxhash.xxh64(e[0]).intdigest() == int(man[1][i])) for i, e in enumerate(kept))
    gathered_ok = all(len(res[i][0]) == int(man[0][i]) and xxhash.xxh64(res[i][0]).intdigest() == int(man[1][i]) for i in range(0, n_files, max(1, n_files // 16)))
    del kept
    doc, binary = gltf.read_glb(glbs[n_files // 2])
    mesh, _ = gltf.primitive_to_mesh(doc, binary, doc["meshes"][0]["primitives"][0])
    return {"workload": f"BASELINE configs[3]: {n_files} GLB files in memory through gltf.transcode_files over {world} rank(s): files dealt by size before anything is parsed, "
                        "each rank transcodes its own files inside the library, finished files gathered on rank 0", "scaling": "strong", "n_gpus": world,
            "triangles": int(total), "ms_per_step": round(dt / steps * 1e3, 2), "value": round(total * steps / dt / 1e6, 2), "unit": "Mtriangles/s",
            "files_owned_per_rank_min_max": [own_min, own_max], "files_on_rank0": len(res),
            "own_files_ms_max_over_ranks": round(o_max * 1e3, 2), "gather_ms_max_over_ranks": round(g_max * 1e3, 2),
            "files_stay_on_their_ranks": {"what": "gather=\"manifest\": every rank keeps (writes) the files it made; sizes + xxh64 digests of all files on every rank (one all_reduce of 16 bytes per file)",
                                          "ms_per_step": round(dt_m / steps * 1e3, 2), "value": round(total * steps / dt_m / 1e6, 2), "unit": "Mtriangles/s",
                                          "manifest_matches_the_gathered_files": bool(mine_ok and gathered_ok)},
            "sample_blob_equals_whole_mesh_encode": bool(bytes(res[n_files // 2][1][0]) == dmi.encode_mesh(mesh, cfg))}


def transcode_one_process(n_files, devices, steps=3):
    """BASELINE configs[3] through ONE process and a list of devices: dmi_transcode_assets deals the FILES to the devices by size before anything is parsed
    (LPT), parses them on a small pool of threads, one dmi_transcoder per device (stage = a quarter of the device's share), the library's threads write the
    files.  No JSON parsed N times, no padded gather.  Strong (n_files in all) and weak (n_files per device: the same list N times over)."""
    from draco_oxide_amd import binding
    glbs, total = synth.batch_glbs(n_files)
    share = os.environ.get("DMI_HOST_THREADS")          # (the ranks of this run split the host's CPUs among them; this ONE process drives all devices: it gets them all —
    os.environ["DMI_HOST_THREADS"] = str(usable_cpus())  #  the other ranks wait at a barrier meanwhile)
    out = {"devices": list(devices)}
    try:
        for form, lst, tris in (("strong", glbs, total), ("weak", glbs * len(devices), total * len(devices))):
            if form == "weak" and len(devices) == 1:
                continue
            al = binding.AssetList(lst)
            binding.transcode_assets(al, devices=devices)
            ts, sts = [], []
            for _ in range(steps):
                t0 = time.perf_counter()
                res, st = binding.transcode_assets(al, devices=devices)
                ts.append(time.perf_counter() - t0)
                sts.append(st)
                del res
            med, best = _med_min(ts)
            st = sts[ts.index(sorted(ts)[len(ts) // 2])]
            out[form] = {"files": len(lst), "triangles": int(tris), "value": round(tris / med / 1e6, 2), "unit": "Mtriangles/s", "statistic": f"median of {steps} calls",
                         "ms_per_batch": round(med * 1e3, 2), "ms_per_batch_min": round(best * 1e3, 2), "parse_ms": round(st["parse_ms"], 2), "parse_threads": int(st["parse_threads"]),
                         "pushed_ms": round(st["pushed_ms"], 2), "last_stage_coded_at_ms": round(st["finished_ms"], 2), "stages_per_device": round(st["stages"] / max(1, len(devices)), 2),
                         "primitives_built_by_the_host_builder": int(st["primitives_host_built"])}
    finally:
        if share is None:
            os.environ.pop("DMI_HOST_THREADS", None)
        else:
            os.environ["DMI_HOST_THREADS"] = share
    out.update({"triangles": out["strong"]["triangles"], "value": out["strong"]["value"], "unit": "Mtriangles/s", "scaling": "strong (see `weak` beside it)",
                "ms_per_batch": out["strong"]["ms_per_batch"]})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=2236, help="grid side n (F = 2 n^2); default = the 10M-triangle workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch", action="store_true", help="skip the extra batch-regime measurements (outside the timed steps)")
    ap.add_argument("--no-scopes", action="store_true", help="skip the sub-scope measurements (N=1 only, outside the timed steps)")
    ap.add_argument("--batch-meshes", type=int, default=1024, help="size of the sharded batch at N > 1")
    ap.add_argument("--transcode-files", type=int, default=1024, help="GLB files of the transcode regime (BASELINE configs[3]); 0 = skip")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child runs that measure roofline.traffic (N=1 only)")
    args = ap.parse_args()
    # roofline.traffic of THIS run: child processes under rocprofv3, before this process touches the GPU (N = 1 only; DMI_ROOFLINE_TRAFFIC: a profiling session
    # — scripts/profile_round.sh — hands its own counters over instead)
    measured_traffic = None
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_traffic and not os.environ.get("DMI_ROOFLINE_TRAFFIC"):
        measured_traffic = measure_traffic(args.grid)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available() or dmi.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: libdraco_mi has no CPU fallback")
    # a process dedicated to encoding opts in to what the library no longer does to its host by default (round 6, dmi_configure_process): its large index
    # arrays on huge pages, the calling thread kept on the GPU's memory node during whole-mesh calls (DMI_BENCH_PLAIN_PROCESS=1: both off, as the tests run)
    plain_process = os.environ.get("DMI_BENCH_PLAIN_PROCESS") is not None
    if not plain_process:
        dmi.configure_process(huge_page_new=True, numa_pin=True)
    # DMI_BENCH_BACKEND=gloo lets the N>1 control flow — and the host-side contention of N ranks on one box — be measured on a 1-GPU box
    # (all ranks share cuda:0, the gather runs on CPU tensors); the driver's multi-GPU runs use the default: nccl = RCCL over xGMI.
    backend = os.environ.get("DMI_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    gather_dev = dev if backend == "nccl" else torch.device("cpu")
    host_threads = usable_cpus()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the ranks share the host: each one's library threads (connectivity walks, splice, host-core chains) get an equal share
        os.environ.setdefault("DMI_HOST_THREADS", str(max(2, usable_cpus() // world)))
        host_threads = int(os.environ["DMI_HOST_THREADS"])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # every rank owns one mesh of the workload shape (weak scaling); different seeds → different meshes
    mesh = synth.torus_mesh(args.grid, seed=synth.SEED + rank)
    n_tris = len(mesh.faces)
    dmesh = dmi.DeviceMesh.upload(mesh, local_rank)      # faces + attribute values resident in HBM before the timed region
    cmesh = dmesh._c()
    # the job of every step launches on a torch-owned HIP stream; per-stage times come from hipEvents the library records on it
    tstream = torch.cuda.Stream(dev)
    cfg = dmi.Config(device=local_rank, stream=tstream.cuda_stream, flags=dmi.FLAG_TIMINGS)

    gather_s = [0.0]

    def step():
        # the product boundary is the C ABI: the `.drc` lands in a library-owned host buffer, which the gather at N > 1 reads in place
        with dmi.encode_mesh_device_raw(dmesh, cfg, cmesh) as out:
            if world > 1:
                tg = time.perf_counter()
                dmi_dist.gather_bitstreams(out.view(0), device=gather_dev, as_bytes=False)
                gather_s[0] += time.perf_counter() - tg
            return out.nbytes

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    gather_s[0] = 0.0
    keys = ("quantize_ms", "predict_ms", "histogram_ms", "table_ms", "rans_ms", "total_ms", "longest_stream_ms", "readback_wait_ms",
            "mesh_readback_ms", "tables_ms", "connectivity_ms", "job_create_ms", "call_ms", "job_create_device_ms", "early_ms")
    stages = {k: 0.0 for k in keys}
    t_start = time.perf_counter()
    out_len = 0
    for _ in range(args.steps):
        out_len = step()
        tm = dmi.last_call_timings()
        for k in stages:
            stages[k] += tm[k]
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tm = dmi.last_call_timings()
    for k in stages:
        stages[k] /= max(args.steps, 1)
    step_s = elapsed / max(args.steps, 1)
    dist_info = None
    if world > 1:
        # what the collective layer really was in this run: backend, world size, per-rank gather time and usable host CPUs
        mine = torch.tensor([gather_s[0] / max(args.steps, 1) * 1e3, float(usable_cpus()), float(host_threads), float(local_rank)], dtype=torch.float64, device=gather_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                     "gather_ms_per_step_by_rank": [round(float(t[0]), 3) for t in allr], "host_cpus_usable_by_rank": [int(t[1]) for t in allr],
                     "library_host_threads_by_rank": [int(t[2]) for t in allr], "hip_device_by_rank": [int(t[3]) for t in allr]}

    line = None
    if rank == 0:
        total_tris = n_tris * world * args.steps
        value = total_tris / elapsed / 1e6
        # everything that quantizes and predicts: the early stage (value ranges + value-order quantization, issued behind the device stage's read-backs,
        # alone on the device while the host walks: its hipEvent span is its kernel time) + the launches after the walks (sweep, fix-up); SURVEY §8d's bytes
        # cover both, so both are in the denominator (the round-5 line left the early stage out)
        after_ms = stages["quantize_ms"] + stages["predict_ms"]
        pass_ms = after_ms + stages["early_ms"]
        achieved = tm["predict_bytes"] / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0
        longest_symbols = n_tris // 2 * 3   # the position stream: V·3 symbols
        hybrid = bool(tm["host_chains"])
        gpu_ms = stages["tables_ms"] + stages["quantize_ms"] + stages["predict_ms"] + stages["histogram_ms"] + stages["table_ms"]
        traffic = os.environ.get("DMI_ROOFLINE_TRAFFIC") or measured_traffic   # a profiling session's counters (scripts/profile_round.sh), else this run's own child runs; never read from a file
        line = {
            "metric": "Mtriangles/sec encoded (bit-exact .drc) at 1/2/4/8 MI355X vs CPU ref",
            "value": round(value, 3), "unit": "Mtriangles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_s * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: {n_tris}-triangle synthetic torus grid (n={args.grid}) per GPU, pos+normals+UV, "
                                   "Edgebreaker order, parallelogram/normal/texcoord prediction, 11/8/10-bit (encode::Config::default()); "
                                   "one step = dmi_encode_mesh_device = encode::encode(mesh): mesh (faces + attribute values) resident in HBM in, whole .drc in a host buffer out — "
                                   "device corner tables, the two serial host walks (Edgebreaker traversal, sequencer), relabelling, the attribute-encoding hot path, splice; "
                                   "bytes identical to the oracle's (tests: 10M-triangle byte parity; reference-made .drc files do not exist: parity is oracle-exact)",
                       "triangles_per_gpu": n_tris, "attributes": "pos3+nrm3+uv2", "bitstream_bytes": out_len,
                       "parallelism": f"{world} independent meshes, one per GPU" + (", RCCL gather of bitstreams to rank 0" if world > 1 else ""),
                       "host_threads_per_rank": host_threads,
                       "process_options": "none (library defaults)" if plain_process else "dmi_configure_process: huge-page operator new + NUMA pin (opt-in)"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                         "frac_of_achievable": round(achieved / HBM_ACHIEVABLE_GBPS, 5), "achievable_gbps": HBM_ACHIEVABLE_GBPS,
                         "traffic": int(traffic) if traffic else None,
                         "traffic_over_algorithmic": round(int(traffic) / max(int(tm["predict_bytes"]), 1), 3) if traffic else None,
                         "traffic_profiled": profiled_traffic(),
                         "kernel": ("quantize+predict pass of a call whose values are in HBM when it starts = EVERY launch that quantizes or predicts: the early stage (value ranges, value-order "
                                    "quantization into 16-byte records + the quantized values' min/max partials; issued behind the device stage's last read-back, so it runs alone during the host walks) "
                                    "+ every launch of the job's stream between the end of the host walks and the histogram stage (coding-order gather of the records, fused predictor sweep, fix-up of deferred entries)"
                                    if stages["early_ms"] > 0 else
                                    "quantize+predict pass = every launch between the first kernel and the histogram stage of one step "
                                    "(value ranges, coding-order gather + quantize, min/max finals, fused predictor sweep)") + ", hipEvent-timed on the streams the kernels launch on, inside the timed steps",
                         "algorithmic_bytes": int(tm["predict_bytes"]), "duration_ms": round(pass_ms, 4),
                         "early_stage_ms": round(stages["early_ms"], 4), "after_walks_ms": round(after_ms, 4),
                         # the same bytes over everything the device does per encode::encode call to run the pass: job creation (coding-order relabelling of the
                         # tables, map compositions, fan rows, buffer clears: its device span on the job's stream) + the early stage + the pass
                         "call_inclusive": {"duration_ms": round(pass_ms + stages["job_create_device_ms"], 4), "job_create_device_ms": round(stages["job_create_device_ms"], 4),
                                            "achieved": round(tm["predict_bytes"] / max((pass_ms + stages["job_create_device_ms"]) * 1e-3, 1e-12) / 1e9, 2),
                                            "frac": round(tm["predict_bytes"] / max((pass_ms + stages["job_create_device_ms"]) * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBPS, 5)}},
            "stages_ms": {k: round(v, 4) for k, v in stages.items()},
            "step_split_ms": {"mesh_readback (faces → host for the walks: issue only — the copy runs beside the table kernels)": round(stages["mesh_readback_ms"], 3),
                              "device corner tables + read-back (incl. the faces' copy)": round(stages["tables_ms"], 3),
                              "host walks: Edgebreaker traversal + connectivity bytes + sequencer": round(stages["connectivity_ms"] - stages["tables_ms"], 3),
                              "job creation: relabelling, fan rows, buffers": round(stages["job_create_ms"], 3),
                              "attribute-encoding hot path + splice (dmi_job_encode)": round(stages["total_ms"], 3)},
            "gpu_time_fraction_of_step": round(gpu_ms / max(stages["call_ms"], 1e-9), 4),
            "host_serial_fraction_of_step": round((stages["connectivity_ms"] - stages["tables_ms"] + stages["rans_ms"]) / max(stages["call_ms"], 1e-9), 4),
            "chains": {"form": "hybrid: symbols + device-built tables read back, one host core per stream" if hybrid else "device: scalar-unit walker + emitter wavefronts",
                       "streams": int(tm["num_streams"]), "symbols": int(tm["symbols"]), "longest_stream_symbols": longest_symbols},
        }
        try:   # the fraction against a device copy measured on this box as well as against the nominal peak
            n_copy = 1 << 28   # 1 GiB of f32 read + 1 GiB written per copy
            src = torch.empty(n_copy, dtype=torch.float32, device=dev).fill_(1.0)
            dst = torch.empty_like(src)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            dst.copy_(src); torch.cuda.synchronize(dev)
            e0.record()
            for _ in range(10):
                dst.copy_(src)
            e1.record(); torch.cuda.synchronize(dev)
            copy_gbps = 10 * 2 * n_copy * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
            line["roofline"]["measured_torch_copy_gbps"] = round(copy_gbps, 1)
            del src, dst
        except Exception as e:
            line["roofline"]["measured_copy_error"] = str(e)[:120]
        if hybrid and stages["longest_stream_ms"] > 0:
            line["chains"]["host_core_msym_per_s"] = round(longest_symbols / stages["longest_stream_ms"] / 1e3, 2)

    if world == 1 and not args.no_scopes:
        # ---- the other scopes of the same workload (outside the timed steps) ----
        try:
            conn = dmi.encode_connectivity(mesh)             # host builders (no GPU): the tables the boundary call takes
            tables = [conn.table(i) for i in range(conn.num_tables)]
            seeds = conn.seeds()
            job = dmi.Job.from_tables(mesh.attributes, tables, seeds=seeds, cfg=dmi.Config(device=local_rank, flags=dmi.FLAG_TIMINGS))
            job.encode_raw().free()
            tr = []
            for _ in range(5):
                t0 = time.perf_counter()
                job.encode_raw().free()
                tr.append(time.perf_counter() - t0)
            rt = job.timings()
            line["resident_attribute_step"] = {"call": "dmi_job_encode on a resident job: the attribute-encoding hot path alone, connectivity outputs in HBM (rounds 1–2 reported this as `value`)",
                                               "ms": round(min(tr) * 1e3, 3), "mtri_per_s": round(n_tris / min(tr) / 1e6, 2),
                                               "gpu_stages_ms": round(rt["quantize_ms"] + rt["predict_ms"] + rt["histogram_ms"] + rt["table_ms"], 4), "host_chains_ms": round(rt["rans_ms"], 3)}
            # the same streams on the device walker (DMI_CHAINS is read at job creation): what the hybrid form replaces
            os.environ["DMI_CHAINS"] = "device"
            dj = dmi.Job.from_tables(mesh.attributes, tables, seeds=seeds, cfg=dmi.Config(device=local_rank, flags=dmi.FLAG_TIMINGS))
            del os.environ["DMI_CHAINS"]
            ref_bytes = dj.encode()
            t0 = time.perf_counter()
            dj.encode_raw().free()
            dev_s = time.perf_counter() - t0
            dtm = dj.timings()
            dj.close()
            with job.encode_raw() as o:
                same = o[0] == ref_bytes
            line["chains"].update({"device_walker_msym_per_s": round((n_tris // 2 * 3) / max(dtm["rans_ms"], 1e-9) / 1e3, 2),
                                   "device_only_attribute_step_ms": round(dev_s * 1e3, 2), "device_only_attribute_step_mtri_per_s": round(n_tris / dev_s / 1e6, 2),
                                   "forms_byte_identical": bool(same)})
            job.close()
            dmi.encode_attributes(mesh.attributes, tables, seeds=seeds, cfg=dmi.Config(device=local_rank))   # warm the pinned pools
            tb = []
            for _ in range(2):
                t0 = time.perf_counter()
                dmi.encode_attributes(mesh.attributes, tables, seeds=seeds, cfg=dmi.Config(device=local_rank))
                tb.append(time.perf_counter() - t0)
            line["boundary_call"] = {"call": "dmi_encode_attributes: host pointers in (attributes, corner tables, sequences), attribute-section bytes out — "
                                             "uploads, coding-order relabelling, encode, read-back (the call the Rust shim binds)",
                                     "seconds": round(min(tb), 4), "mtri_per_s": round(n_tris / min(tb) / 1e6, 2)}
            conn.close()
            te = []
            for _ in range(3):
                t0 = time.perf_counter()
                drc = dmi.encode_mesh(mesh, dmi.Config(device=local_rank))
                te.append(time.perf_counter() - t0)
            line["end_to_end_host_memory"] = {"call": "dmi_encode_mesh: the same encode with the mesh in HOST memory (PCIe-inclusive: faces and values go up first)",
                                              "seconds": round(min(te), 4), "mtri_per_s": round(n_tris / min(te) / 1e6, 2), "drc_bytes": len(drc)}
            os.environ["DMI_HOST_CONNECTIVITY"] = "1"
            t0 = time.perf_counter()
            drc2 = dmi.encode_mesh(mesh, dmi.Config(device=local_rank))
            t_host = time.perf_counter() - t0
            del os.environ["DMI_HOST_CONNECTIVITY"]
            line["end_to_end_host_memory"].update({"host_tables_form_seconds": round(t_host, 4), "host_tables_form_same_bytes": drc2 == drc})
        except Exception as e:
            os.environ.pop("DMI_CHAINS", None)
            os.environ.pop("DMI_HOST_CONNECTIVITY", None)
            line["scopes_error"] = str(e)[:200]

    if not args.no_batch:
        if world == 1:   # reported beside the headline, never part of `value`
            try:
                line["batch_regime"] = batch_regime(device=local_rank)
            except Exception as e:   # the headline line must not depend on it
                line["batch_regime"] = {"error": str(e)[:200]}
            if args.transcode_files > 0:
                try:
                    line["transcode_regime"] = transcode_regime(args.transcode_files, device=local_rank)
                except Exception as e:
                    line["transcode_regime"] = {"error": str(e)[:200]}
        else:
            try:
                res = batch_sharded(args.batch_meshes, rank, world, local_rank, gather_dev)
                if rank == 0:
                    line["batch_sharded"] = res
            except Exception as e:
                if rank == 0:
                    line["batch_sharded"] = {"error": str(e)[:200]}
            if args.transcode_files > 0:
                try:
                    res = transcode_sharded(args.transcode_files, rank, world, local_rank, gather_dev)
                    if rank == 0:
                        line["transcode_sharded"] = res
                except Exception as e:
                    if rank == 0:
                        line["transcode_sharded"] = {"error": str(e)[:200]}
                # the same list through ONE process driving all N devices (dmi_transcode_assets with a device list: one dmi_transcoder per GPU, no second
                # interpreter, no gather): rank 0 runs it while the other ranks wait at the barrier below
                if rank == 0:   # (on a 1-GPU box — scripts/scaling_one_host.sh — the N transcoders share cuda:0)
                    try:
                        line["transcode_one_process_n_devices"] = transcode_one_process(args.transcode_files, [k % max(torch.cuda.device_count(), 1) for k in range(world)])
                    except Exception as e:
                        line["transcode_one_process_n_devices"] = {"error": str(e)[:200]}
                dist.barrier()
    if rank == 0:
        if dist_info is not None:
            line["distributed"] = dist_info
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(mesh)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
