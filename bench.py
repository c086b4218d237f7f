#!/usr/bin/env python3
"""bench.py — attribute-encoding hot path of draco-oxide on MI355X.

One "step" = one pass of the hot path (value ranges → coding-order gather + quantize → predict + transform → histograms → table
stage → rANS/rABS stream coding → spliced attribute-section bytes on the host) over one resident mesh: `dmi_job_encode`.  Inputs
(raw attributes, corner tables, Edgebreaker-order sequences) are resident in HBM before the timed region starts.  At N > 1 every
rank encodes its own mesh (independent meshes shard with no data-path collective) and the finished bitstreams are gathered onto
rank 0 over RCCL inside the timed region (weak scaling); rank 0 additionally reports `batch_sharded` = BASELINE configs[3] dealt
over the N ranks (strong scaling, outside the timed steps).

Workload (BASELINE.json configs[2], the configuration the metric's target is quoted on): 10M-triangle synthetic closed torus grid
(n=2236 → 9 999 392 triangles, 4 999 696 vertices), positions + normals + UVs, Edgebreaker order, parallelogram / normal /
texcoord prediction, wrapped-difference + octahedral transforms, 11/8/10-bit quantization — `encode::Config::default()`.

Scopes on the line: `value` = the resident hot path; `boundary_call` = `dmi_encode_attributes` with host pointers in
(what the Rust shim binds: uploads + coding-order relabelling + encode + read-back); `end_to_end` = `dmi_encode_mesh`, mesh in →
whole `.drc` out (host Edgebreaker connectivity included).  `cpu_baseline` = the oracle (CPU restatement of the reference, one
core) on the same mesh with its per-stage split.

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

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ≈6.3 TB/s achievable)


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


def cpu_baseline(mesh):
    """Reference algorithm on one host core: the oracle (CPU restatement, kind "port"), timed on the SAME mesh, with the per-stage
    split of BASELINE.md §3; scope of `value` matched to the GPU timed region (attribute section minus the sequencer).  The
    reference's own complexity (`faithful`: linear `contains` scans, O(V²)) is timed on a bounded ≈100k-triangle sample."""
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
        "value": round(f / scope_s / 1e6, 4), "unit": "Mtriangles/s", "cores": 1, "kind": "port", "host_cpus_usable": usable_cpus(),
        "sample": f"whole workload mesh ({f} triangles), oracle `ranked` mode (same bytes, O(1) already-coded test), attribute section minus sequencer = {scope_s:.2f} s; "
                  f"whole .drc {wall:.2f} s = {f / wall / 1e6:.3f} Mtri/s end to end",
        "end_to_end_mtri_per_s": round(f / wall / 1e6, 4),
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


def batch_regime(n_meshes=256, steps=3):
    """The batch form of the same path (BASELINE configs[3] shape): n independent meshes, F log-uniform in [2k, 200k], pos+nrm+uv,
    resident jobs, ONE dmi_jobs_encode per step, timed at the C ABI (the call + dmi_free_many of its outputs)."""
    meshes = synth.batch_meshes(n_meshes)
    total = sum(len(m.faces) for m in meshes)
    t0 = time.time()
    jobs = dmi.meshes_prepare(meshes, dmi.Config())
    prepare_s = time.time() - t0
    with dmi.jobs_encode_raw(jobs):   # warm-up
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        with dmi.jobs_encode_raw(jobs) as batch:
            pass
    dt = (time.perf_counter() - t0) / steps
    with dmi.jobs_encode_raw(jobs) as batch:
        nbytes = batch.nbytes
        assert batch[0] == jobs[0].encode() and batch[n_meshes - 1] == jobs[n_meshes - 1].encode()
    for j in jobs:
        j.close()
    return {"workload": f"{n_meshes} independent meshes, F log-uniform [2k,200k], pos+nrm+uv, one dmi_jobs_encode per step", "triangles": int(total),
            "ms_per_batch": round(dt * 1e3, 3), "value": round(total / dt / 1e6, 2), "unit": "Mtriangles/s", "bitstream_bytes": int(nbytes),
            "host_prepare_s": round(prepare_s, 3), "end_to_end_mtri_per_s": round(total / (prepare_s + dt) / 1e6, 2)}


def batch_sharded(n_meshes, rank, world, local_rank, gather_dev, steps=2):
    """BASELINE configs[3] over the N ranks of this run: the batch dealt by triangle count (LPT), ONE dmi_jobs_encode per rank and
    step on resident jobs, the finished `.drc` blobs gathered onto rank 0 in mesh order (RCCL).  Strong scaling; rank 0 checks a
    sample of the gathered blobs byte for byte against single-job encodes of the same meshes."""
    meshes = synth.batch_meshes(n_meshes)
    weights = [len(m.faces) for m in meshes]
    mine = dmi_dist.shard_indices(n_meshes, rank, world, weights=weights)
    t0 = time.time()
    jobs = dmi.meshes_prepare([meshes[i] for i in mine], dmi.Config(device=local_rank)) if mine else []
    prepare_s = time.time() - t0
    heads = [j.header_and_connectivity for j in jobs]

    def step():
        blobs = []
        if jobs:
            with dmi.jobs_encode_raw(jobs) as out:
                blobs = [h + out[k] for k, h in enumerate(heads)]
        return dmi_dist.gather_blob_lists(blobs, mine, n_meshes, device=gather_dev)

    step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        got = step()
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt, prepare_s], dtype=torch.float64, device=gather_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt, prepare_max = float(t[0].item()), float(t[1].item())
    res = None
    if rank == 0:
        total = sum(weights)
        check = sorted(set([int(np.argmax(weights)), 0, n_meshes - 1] + list(range(0, n_meshes, max(1, n_meshes // 16)))))
        ok = True
        for i in check:
            ok = ok and bytes(got[i]) == dmi.encode_mesh(meshes[i], dmi.Config(device=local_rank))
        res = {"workload": f"BASELINE configs[3]: {n_meshes} meshes, F log-uniform [2k,200k], pos+nrm+uv, dealt over {world} rank(s) by triangle count (LPT), "
                           "one dmi_jobs_encode per rank and step, blobs gathered on rank 0", "scaling": "strong", "n_gpus": world, "triangles": int(total),
               "ms_per_step": round(dt / steps * 1e3, 3), "value": round(total * steps / dt / 1e6, 2), "unit": "Mtriangles/s",
               "host_prepare_s_max_over_ranks": round(prepare_max, 3), "blobs_on_rank0": len(got), "sample_checked_against_single_encodes": len(check), "sample_ok": bool(ok)}
    for j in jobs:
        j.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=2236, help="grid side n (F = 2 n^2); default = the 10M-triangle workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch", action="store_true", help="skip the extra batch-regime measurements (outside the timed steps)")
    ap.add_argument("--no-scopes", action="store_true", help="skip boundary_call / end_to_end / device-chain comparison (N=1 only, outside the timed steps)")
    ap.add_argument("--batch-meshes", type=int, default=1024, help="size of the sharded batch at N > 1")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available() or dmi.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: libdraco_mi has no CPU fallback")
    # DMI_BENCH_BACKEND=gloo lets the N>1 control flow be exercised on a 1-GPU box (all ranks share cuda:0,
    # the gather runs on CPU tensors); the driver's multi-GPU runs use the default: nccl = RCCL over xGMI.
    backend = os.environ.get("DMI_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    gather_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the ranks share the host: each one's library threads (connectivity walks, splice, host-core chains) get an equal share
        os.environ.setdefault("DMI_HOST_THREADS", str(max(2, usable_cpus() // world)))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # every rank owns one mesh of the workload shape (weak scaling); different seeds → different meshes
    mesh = synth.torus_mesh(args.grid, seed=synth.SEED + rank)
    n_tris = len(mesh.faces)
    # the job launches on a torch-owned HIP stream; per-stage times come from hipEvents recorded on it
    tstream = torch.cuda.Stream(dev)
    stream = tstream.cuda_stream
    t0 = time.time()
    conn = dmi.encode_connectivity(mesh)             # host: corner tables, Edgebreaker, sequencers (the reference's connectivity stage)
    connectivity_s = time.time() - t0
    tables = [conn.table(i) for i in range(conn.num_tables)]
    seeds = conn.seeds()
    t0 = time.time()
    job = dmi.Job.from_tables(mesh.attributes, tables, seeds=seeds, cfg=dmi.Config(device=local_rank, stream=stream, flags=dmi.FLAG_TIMINGS))
    job_create_s = time.time() - t0

    def step():
        # the product boundary is the C ABI: the section lands in a library-owned host buffer, which the gather at N > 1 reads in place
        # (rank 0 receives every rank's section in one pinned host buffer)
        with job.encode_raw() as out:
            if world > 1:
                dmi_dist.gather_bitstreams(out.view(0), device=gather_dev, as_bytes=False)
            return out.nbytes

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    keys = ("quantize_ms", "predict_ms", "histogram_ms", "table_ms", "rans_ms", "total_ms", "longest_stream_ms", "readback_wait_ms")
    stages = {k: 0.0 for k in keys}
    t_start = time.perf_counter()
    out_len = 0
    for _ in range(args.steps):
        out_len = step()
        tm = job.timings()
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
    tm = job.timings()
    for k in stages:
        stages[k] /= max(args.steps, 1)
    resident_s = elapsed / max(args.steps, 1)

    line = None
    if rank == 0:
        total_tris = n_tris * world * args.steps
        value = total_tris / elapsed / 1e6
        pass_ms = stages["quantize_ms"] + stages["predict_ms"]
        achieved = tm["predict_bytes"] / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0
        longest_symbols = n_tris // 2 * 3   # the position stream: V·3 symbols
        hybrid = bool(tm["host_chains"])
        traffic = os.environ.get("DMI_ROOFLINE_TRAFFIC")   # set by scripts/profile_round.sh from the rocprofv3 PMC passes of the same session; never read from a file
        line = {
            "metric": "Mtriangles/sec encoded (bit-exact .drc) at 1/2/4/8 MI355X vs CPU ref",
            "value": round(value, 3), "unit": "Mtriangles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(resident_s * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: {n_tris}-triangle synthetic torus grid (n={args.grid}) per GPU, pos+normals+UV, "
                                   "Edgebreaker order, parallelogram/normal/texcoord prediction, 11/8/10-bit (encode::Config::default()); "
                                   "attribute-encoding hot path (encode_attributes = dmi_job_encode) with the connectivity stage's outputs resident in HBM; "
                                   "bytes identical to the oracle's (tests: 10M-triangle byte parity)",
                       "triangles_per_gpu": n_tris, "attributes": "pos3+nrm3+uv2", "bitstream_bytes": out_len,
                       "parallelism": f"{world} independent meshes, one per GPU" + (", RCCL gather of bitstreams to rank 0" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                         "traffic": int(traffic) if traffic else None,
                         "kernel": "quantize+predict pass = every launch between the first kernel and the histogram stage of one step "
                                   "(value ranges, coding-order gather + quantize, min/max finals, fused predictor sweep), hipEvent-timed on the job's stream",
                         "algorithmic_bytes": int(tm["predict_bytes"]), "duration_ms": round(pass_ms, 4)},
            "stages_ms": {k: round(v, 4) for k, v in stages.items()},
            "chains": {"form": "hybrid: symbols + device-built tables read back, one host core per stream" if hybrid else "device: scalar-unit walker + emitter wavefronts",
                       "streams": int(tm["num_streams"]), "symbols": int(tm["symbols"]), "longest_stream_symbols": longest_symbols},
            "host_connectivity_s": round(connectivity_s, 3), "job_create_s": round(job_create_s, 3),
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
            line["roofline"]["measured_copy_gbps"] = round(copy_gbps, 1)
            line["roofline"]["frac_of_measured_copy"] = round(achieved / copy_gbps, 5)
            del src, dst
        except Exception as e:
            line["roofline"]["measured_copy_error"] = str(e)[:120]
        if hybrid and stages["longest_stream_ms"] > 0:
            line["chains"]["host_core_msym_per_s"] = round(longest_symbols / stages["longest_stream_ms"] / 1e3, 2)
        else:
            line["chains"]["device_walker_msym_per_s"] = round(longest_symbols / max(stages["rans_ms"], 1e-9) / 1e3, 2)

    if world == 1 and not args.no_scopes:
        # ---- the other scopes of the same workload (outside the timed steps) ----
        try:
            # the same stream on the device walker (DMI_CHAINS is read at job creation): what the hybrid form replaces
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
                                   "device_form_ms_per_step": round(dev_s * 1e3, 2), "forms_byte_identical": bool(same)})
        except Exception as e:
            os.environ.pop("DMI_CHAINS", None)
            line["chains"]["device_form_error"] = str(e)[:200]
        try:
            dmi.encode_attributes(mesh.attributes, tables, seeds=seeds, cfg=dmi.Config(device=local_rank))   # warm the pinned pools
            tb = []
            for _ in range(2):
                t0 = time.perf_counter()
                dmi.encode_attributes(mesh.attributes, tables, seeds=seeds, cfg=dmi.Config(device=local_rank))
                tb.append(time.perf_counter() - t0)
            line["boundary_call"] = {"call": "dmi_encode_attributes: host pointers in (attributes, corner tables, sequences), attribute-section bytes out — "
                                             "uploads, coding-order relabelling, encode, read-back (the call the Rust shim binds)",
                                     "seconds": round(min(tb), 4), "mtri_per_s": round(n_tris / min(tb) / 1e6, 2)}
            te = []
            for _ in range(3):
                t0 = time.perf_counter()
                drc = dmi.encode_mesh(mesh, dmi.Config(device=local_rank))
                te.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            dmi.encode_connectivity(mesh).close()
            conn_warm_s = time.perf_counter() - t0
            line["end_to_end"] = {"call": "dmi_encode_mesh: mesh in, whole .drc out (host corner tables + Edgebreaker + sequencers, uploads, device attribute section, splice); "
                                          "`seconds` = a call of a running process (the library recycles its large host arrays between calls), `first_call_seconds` = the "
                                          "first one (every array freshly mapped)",
                                  "seconds": round(min(te[1:]), 4), "mtri_per_s": round(n_tris / min(te[1:]) / 1e6, 2), "first_call_seconds": round(te[0], 4),
                                  "drc_bytes": len(drc), "host_connectivity_s": round(conn_warm_s, 3), "host_connectivity_first_call_s": round(connectivity_s, 3),
                                  "job_create_s": round(job_create_s, 3), "encode_s": round(resident_s, 4)}
        except Exception as e:
            line["scopes_error"] = str(e)[:200]
    conn.close()
    job.close()

    if not args.no_batch:
        if world == 1:   # reported beside the headline, never part of `value`
            try:
                line["batch_regime"] = batch_regime()
            except Exception as e:   # the headline line must not depend on it
                line["batch_regime"] = {"error": str(e)[:200]}
        else:
            try:
                res = batch_sharded(args.batch_meshes, rank, world, local_rank, gather_dev)
                if rank == 0:
                    line["batch_sharded"] = res
            except Exception as e:
                if rank == 0:
                    line["batch_sharded"] = {"error": str(e)[:200]}
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(mesh)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
