"""BASELINE configs[3] on N GPUs: a batch of independent meshes dealt to the ranks by triangle count (LPT), ONE dmi_jobs_encode per rank
and step on resident jobs, the finished sections gathered onto rank 0 in mesh order over RCCL — barrier, K timed steps, barrier, max over
ranks, one JSON line on rank 0.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P scripts/bench_sharded.py [meshes] [steps]
(DMI_BENCH_BACKEND=gloo runs the same control flow with all ranks on cuda:0 and a CPU gather.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import draco_oxide_amd as dmi
from draco_oxide_amd import distributed as dd, synth

n_meshes = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
backend = os.environ.get("DMI_BENCH_BACKEND", "nccl")
local = local if backend == "nccl" else 0
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
gather_dev = dev if backend == "nccl" else torch.device("cpu")
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
meshes = synth.batch_meshes(n_meshes)                       # every rank generates the same list; it only prepares its share
weights = [len(m.faces) for m in meshes]
mine = dd.shard_indices(n_meshes, rank, world, weights=weights)
t0 = time.time()
jobs = dmi.meshes_prepare([meshes[i] for i in mine], dmi.Config(device=local)) if mine else []
prepare_s = time.time() - t0
heads = [j.header_and_connectivity for j in jobs]


def step():
    if jobs:
        with dmi.jobs_encode_raw(jobs) as out:
            blobs = [np.concatenate([np.frombuffer(h, np.uint8), out.view(k)]) for k, h in enumerate(heads)] if world > 1 else None
            nbytes = out.nbytes
    else:
        blobs, nbytes = [], 0
    if world > 1:
        return dd.gather_blob_lists([b.tobytes() for b in blobs], mine, n_meshes, device=gather_dev)
    return nbytes


step()
if world > 1:
    dist.barrier()
torch.cuda.synchronize(dev)
t0 = time.perf_counter()
for _ in range(steps):
    got = step()
torch.cuda.synchronize(dev)
if world > 1:
    dist.barrier()
dt = time.perf_counter() - t0
if world > 1:
    t = torch.tensor([dt], dtype=torch.float64, device=gather_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
if rank == 0:
    total = sum(weights)
    print(json.dumps({"workload": f"{n_meshes} meshes, F log-uniform [2k,200k], pos+nrm+uv, sharded over {world} GPU(s) by triangle count, gathered on rank 0", "n_gpus": world,
                      "triangles": int(total), "ms_per_step": round(dt / steps * 1e3, 3), "value": round(total * steps / dt / 1e6, 2), "unit": "Mtriangles/s",
                      "rank0_share": len(mine), "host_prepare_s_rank0": round(prepare_s, 2), "blobs_on_rank0": (len(got) if world > 1 else len(jobs))}), flush=True)
for j in jobs:
    j.close()
if world > 1:
    dist.destroy_process_group()
