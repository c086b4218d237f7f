"""GPU fuzz at sizes where the large-job forms run by themselves (tile-sorted quantize gather ≥ 2^18 sequence entries, hybrid host-core chains,
device corner tables): grids of 0.5–1.6M triangles, random attribute sets and bit widths, bytes against the oracle; every file read back by
dmi_decode_mesh.  usage: fuzz_gpu_large.py [cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import draco_oxide_amd as dmi
from helpers import oracle_from_product_mesh
from test_gpu_decode import numpy_quantize
from test_gpu_decode_mesh import _canonical_faces, _requantize

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
rng = np.random.default_rng(seed0)
bad = 0
t0 = time.time()
for c in range(n_cases):
    n = int(rng.integers(515, 900))
    ob, nrm, uvs = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    pb, ub = int(rng.integers(8, 17)), int(rng.integers(8, 15))
    mesh = dmi.synth.torus_mesh(n, seed=seed0 + c, open_boundary=ob, normals=nrm, uvs=uvs)
    cfg = dmi.Config(pos_bits=pb, uv_bits=ub)
    want = oracle_from_product_mesh(mesh).encode(pos_bits=pb, uv_bits=ub)
    got = {"whole": dmi.encode_mesh(mesh, cfg), "mesh in HBM": dmi.encode_mesh_device(dmi.DeviceMesh.upload(mesh), cfg)}
    os.environ["DMI_TILE_SORT"] = "0"
    got["no tile sort"] = dmi.encode_mesh(mesh, cfg)
    del os.environ["DMI_TILE_SORT"]
    for name, g in got.items():
        if g != want:
            print(f"case {c}: grid {n} open={ob} nrm={nrm} uv={uvs} {pb}/{ub} bits: {name} differs ({len(g)} vs {len(want)} bytes)"); bad += 1
    dm = dmi.decode_mesh(want)
    pos = mesh.attributes[0]
    q, mn, rg = numpy_quantize(pos.values, pb)
    in_faces = np.asarray(mesh.faces, np.int64).reshape(-1, 3)
    got_rows = _requantize(dm["attributes"][0]["values"], mn, rg, pb)[dm["faces"].astype(np.int64)]
    if not (dm["faces"].shape == in_faces.shape and (_canonical_faces(q[in_faces]) == _canonical_faces(got_rows)).all()):
        print(f"case {c}: decode_mesh does not give the input's triangles"); bad += 1
print(f"{n_cases} large cases in {time.time() - t0:.0f} s, {bad} mismatches")
sys.exit(1 if bad else 0)
