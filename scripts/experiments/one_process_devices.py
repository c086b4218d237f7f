"""dmi_transcode_assets over a device list on one GPU box (every entry cuda:0): python scripts/experiments/one_process_devices.py [n_devices ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from draco_oxide_amd import binding, synth
glbs, total = synth.batch_glbs(1024)
for n in [int(x) for x in sys.argv[1:]] or [1, 2, 4]:
    devices = [0] * n
    binding.transcode_assets(glbs, devices=devices)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); res, st = binding.transcode_assets(glbs, devices=devices); ts.append(time.perf_counter() - t0); del res
    print(f"{n} transcoders on cuda:0: " + " ".join(f"{t * 1e3:.1f}" for t in ts) + " ms", flush=True)
