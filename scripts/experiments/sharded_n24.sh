export DMI_BENCH_BACKEND=gloo
mkdir -p gpurun_out/r6
for n in 2 4; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --steps 3 --warmup 1 --no-cpu-baseline --no-scopes --no-traffic --batch-meshes 256 2>gpurun_out/r6/n$n.err | tail -1 > gpurun_out/r6/n$n.json
  python3 -c "
import json; d=json.loads(open('gpurun_out/r6/n$n.json').read()); t=d.get('transcode_sharded') or {}
print($n, d['value'], d['ms_per_step'], t.get('value'), t.get('ms_per_step'), t.get('own_files_ms_max_over_ranks'), t.get('gather_ms_max_over_ranks'), t.get('files_stay_on_their_ranks'))"
done
