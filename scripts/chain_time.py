#!/usr/bin/env python3
"""The host-core rANS coder on its own: 15M symbols of a geometric source over a 200-symbol alphabet (precision 12), Msym/s.  No device."""
import os, sys, time, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import draco_oxide_amd as dmi
rng = np.random.default_rng(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 15_000_000
sym = np.minimum(rng.geometric(0.06, size=n) - 1, 199).astype(np.uint32)
hist = np.bincount(sym, minlength=200).astype(np.float64)
freq = np.maximum(1, np.floor(hist / hist.sum() * 4096)).astype(np.int64)
freq[np.argmax(freq)] += 4096 - freq.sum()
assert freq.min() >= 1 and freq.sum() == 4096
freq = freq.astype(np.uint32)
best = 1e9
for _ in range(5):
    t = time.perf_counter(); out = dmi.host_rans_stream(freq, 12, sym); best = min(best, time.perf_counter() - t)
print(f"{n / best / 1e6:.1f} Msym/s  ({best * 1e3:.1f} ms, {len(out)} bytes, sha1 {hashlib.sha1(out).hexdigest()[:12]})")
