#!/usr/bin/env python3
"""dmi_transcode_assets over a device list with 1024 files PER entry (the weak form of bench.py's transcode_one_process): python one_process_weak.py <n_devices> [files_per_device]
(DMI_TRACE_STAGES=1 for the stage lines).  On a 1-GPU box every entry is device 0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import draco_oxide_amd as dmi  # noqa: E402
from draco_oxide_amd import binding, synth  # noqa: E402

dmi.configure_process(huge_page_new=True, numa_pin=True)
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 2
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
glbs, total = synth.batch_glbs(per)
al = binding.AssetList(glbs * nd)
devs = [0] * nd
for k in range(4):
    t0 = time.perf_counter()
    res, st = binding.transcode_assets(al, devices=devs)
    dt = time.perf_counter() - t0
    del res
    print(f"{nd} x {per} files on {devs}: {dt * 1e3:.1f} ms = {total * nd / dt / 1e6:.1f} Mtri/s; parse {st['parse_ms']:.1f} pushed {st['pushed_ms']:.1f} finished {st['finished_ms']:.1f} "
          f"build {st['build_ms']:.1f} prepare {st['prepare_ms']:.1f} encode {st['encode_ms']:.1f} assemble {st['assemble_ms']:.1f} stages {st['stages']}", flush=True)
