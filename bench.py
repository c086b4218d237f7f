#!/usr/bin/env python3
"""bench.py — attribute-encoding hot path of draco-oxide on MI355X.

One "step" = one pass of the hot path (quantize → sequence-order gather → predict+transform →
histogram → table normalisation → rANS/rABS coding → spliced attribute-section bytes on the host)
over one resident mesh; at N > 1 every rank encodes its own mesh (independent meshes shard with no
data-path collective) and the finished bitstreams are gathered onto rank 0 over RCCL inside the
timed region.  Inputs (raw attributes, corner tables, Edgebreaker-order sequences) are resident in
HBM before the timed region starts; the serial host graph walks that produce them (corner table,
Edgebreaker, sequencer) are the reference's connectivity stage, outside the hot path.

Workload (BASELINE.json configs[2], the configuration the metric's target is quoted on):
10M-triangle synthetic closed torus grid (n=2236 → 9 999 392 triangles, 4 999 696 vertices),
positions + normals + UVs, Edgebreaker order, parallelogram / normal / texcoord prediction,
wrapped-difference + octahedral transforms, 11/8/10-bit quantization — `encode::Config::default()`.

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


def cpu_baseline(mesh, seconds_budget=30.0):
    """Reference algorithm on one host core: the oracle (CPU restatement, kind "port"), timed on the
    SAME mesh.  Scope matched to the GPU timed region: attribute section minus the sequencer."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers  # test infrastructure: allowed in the cpu_baseline leg only
    sess = helpers.oracle_from_product_mesh(mesh)
    t0 = time.time()
    sess.encode(dump=False)
    wall = time.time() - t0
    conn_s, att_s, seq_s = sess.stage_seconds()
    scope_s = max(att_s - seq_s, 1e-9)
    f = len(mesh.faces)
    return {
        "value": round(f / scope_s / 1e6, 4), "unit": "Mtriangles/s", "cores": 1, "kind": "port",
        "sample": f"whole workload mesh ({f} triangles), oracle ranked mode, attribute section minus sequencer = {scope_s:.2f} s "
                  f"(connectivity {conn_s:.2f} s, sequencer {seq_s:.2f} s, whole .drc {wall:.2f} s → {f / wall / 1e6:.3f} Mtri/s); "
                  "the reference's own O(V^2) `contains` scans would take hours at this size",
    }


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
            "host_prepare_s": round(prepare_s, 2)}


def pmc_traffic_bytes():
    """HBM bytes of one quantize+predict pass from the committed rocprofv3 PMC passes (FETCH_SIZE with the gfx950 x2
    correction + WRITE_SIZE, profiles/round1_pmc_traffic.csv, produced by scripts/summarize_profiles.py); None if absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "round1_pmc_traffic.csv")
    try:
        last = [l for l in open(path) if l.startswith("# quantize+predict pass per step")][-1]
        nums = [float(x) for x in __import__("re").findall(r"([0-9.]+) MB", last)]   # fetch raw, fetch doubled, write
        return int((nums[1] + nums[2]) * 1e6)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=2236, help="grid side n (F = 2 n^2); default = the 10M-triangle workload")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch", action="store_true", help="skip the extra batch-regime measurement (N=1 only, outside the timed steps)")
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
    job = dmi.mesh_prepare(mesh, dmi.Config(device=local_rank, stream=stream, flags=dmi.FLAG_TIMINGS))
    prepare_s = time.time() - t0

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
    stages = {k: 0.0 for k in ("quantize_ms", "predict_ms", "histogram_ms", "table_ms", "rans_ms", "total_ms")}
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

    if rank == 0:
        total_tris = n_tris * world * args.steps
        value = total_tris / elapsed / 1e6
        pass_ms = stages["quantize_ms"] + stages["predict_ms"]
        achieved = tm["predict_bytes"] / (pass_ms * 1e-3) / 1e9 if pass_ms > 0 else 0.0
        line = {
            "metric": "Mtriangles/sec encoded (bit-exact .drc) at 1/2/4/8 MI355X vs CPU ref",
            "value": round(value, 3), "unit": "Mtriangles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: {n_tris}-triangle synthetic torus grid (n={args.grid}) per GPU, pos+normals+UV, "
                                   "Edgebreaker order, parallelogram/normal/texcoord prediction, 11/8/10-bit (encode::Config::default()); "
                                   "attribute-encoding hot path (encode_attributes) with connectivity outputs resident in HBM",
                       "triangles_per_gpu": n_tris, "attributes": "pos3+nrm3+uv2", "bitstream_bytes": out_len,
                       "parallelism": f"{world} independent meshes, one per GPU" + (", RCCL gather of bitstreams to rank 0" if world > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                         "traffic": pmc_traffic_bytes(),
                         "kernel": "quantize+predict pass = k_value_ranges + k_value_ranges_final (incl. slab clear) + k_seq_quantize + k_i32_minmax_final + "
                                   "k_predict_fused (every launch between the first and the histogram stage of one step, hipEvent-timed on the job's stream)",
                         "algorithmic_bytes": int(tm["predict_bytes"]), "duration_ms": round(pass_ms, 4)},
            "stages_ms": {k: round(v, 4) for k, v in stages.items()},
            "chains": {"streams": int(tm["num_streams"]), "symbols": int(tm["symbols"]),
                       "msym_per_s_longest_chain": round((n_tris // 2 * 3) / max(stages["rans_ms"], 1e-9) / 1e3, 2)},
            "host_prepare_s": round(prepare_s, 2),
        }
        if not args.no_batch and world == 1:   # reported beside the headline, never part of `value`
            try:
                line["batch_regime"] = batch_regime()
            except Exception as e:   # the headline line must not depend on it
                line["batch_regime"] = {"error": str(e)[:200]}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(mesh)
        print(json.dumps(line), flush=True)
    job.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
