"""Damaged files through dmi_decode_mesh on the GPU (needs an MI355X): truncations and random byte damage of whole `.drc` files with seams, holes and
several components — an error code or some other mesh, never a crash or a hang.  usage: fuzz_decode_damaged.py [mutations per file]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import draco_oxide_amd as dmi
from test_gpu_decode_mesh import _punched

per_file = int(sys.argv[1]) if len(sys.argv) > 1 else 800
rng = np.random.default_rng(5)
rejected = accepted = 0
for seed in range(6):
    mesh = _punched(20 + 5 * seed, 0.05 + 0.05 * seed, seed, bool(seed & 1), bool(seed & 2))
    good = dmi.encode_mesh(mesh, dmi.Config(pos_bits=int(rng.integers(8, 16)), uv_bits=int(rng.integers(8, 14))))
    dmi.decode_mesh(good)
    for cut in range(0, len(good), max(1, len(good) // 50)):
        try:
            dmi.decode_mesh(good[:cut]); accepted += 1
        except dmi.DracoMiError:
            rejected += 1
    for _ in range(per_file):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        try:
            dmi.decode_mesh(bytes(b)); accepted += 1
        except dmi.DracoMiError:
            rejected += 1
print(f"damaged files: {rejected} rejected with an error code, {accepted} decoded to some mesh, no crash")
