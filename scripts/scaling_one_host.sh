#!/bin/bash
# The N = 1, 2, 4, 8 curve of bench.py predicted on ONE GPU box: N ranks share cuda:0 and the host's CPU quota (DMI_BENCH_BACKEND=gloo:
# the gather runs on CPU tensors).  The step is ≈ 95 % host time (two serial graph walks + the host-core stream coders), so what this
# measures — N ranks contending for the same cores and memory — is what bounds the real N-GPU curve; the GPU stages (≈ 3 % of a step)
# serialise on the one device here, which an N-GPU node does not do.
mkdir -p gpurun_out/r6
out=gpurun_out/r6/scaling_one_host.jsonl
mkdir -p gpurun_out/r6
: > $out
export DMI_BENCH_BACKEND=gloo
for n in 1 2 4 8; do
  if [ $n = 1 ]; then
    timeout 600 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-scopes >> $out 2>> gpurun_out/r6/scaling_one_host.err
  else
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 5 --warmup 2 --no-cpu-baseline --batch-meshes 1024 >> $out 2>> gpurun_out/r6/scaling_one_host.err
  fi
done
python - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r6/scaling_one_host.jsonl") if l.startswith("{")]
base = rows[0]["value"]
print("N  value(Mtri/s)  ms/step  efficiency  host_threads/rank  batch_sharded(Mtri/s)  prepare_ms  encode_ms  transcode_sharded(Mtri/s)  ms  files/rank  one_process strong(Mtri/s, ms, parse_ms, pushed_ms, stages/dev)  weak(Mtri/s, ms)  backend  gather_ms/rank")
for r in rows:
    b = r.get("batch_sharded") or {}
    t = r.get("transcode_sharded") or r.get("transcode_regime") or {}
    o = r.get("transcode_one_process_n_devices") or {}
    st, wk = o.get("strong") or {}, o.get("weak") or {}
    print(r["n_gpus"], r["value"], r["ms_per_step"], round(r["value"] / (base * r["n_gpus"]), 3), r["config"].get("host_threads_per_rank"), b.get("value"), b.get("prepare_ms_max_over_ranks"), b.get("encode_ms_max_over_ranks"),
          t.get("value"), t.get("ms_per_step", t.get("ms_per_batch")), t.get("files_owned_per_rank_min_max", t.get("primitives_built_per_rank_min_max")), ("files stay:", (t.get("files_stay_on_their_ranks") or {}).get("value"), (t.get("files_stay_on_their_ranks") or {}).get("ms_per_step")),
          (st.get("value"), st.get("ms_per_batch"), st.get("parse_ms"), st.get("pushed_ms"), st.get("stages_per_device")), (wk.get("value"), wk.get("ms_per_batch")),
          (r.get("distributed") or {}).get("backend"), (r.get("distributed") or {}).get("gather_ms_per_step_by_rank"))
PY
