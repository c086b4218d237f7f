#!/bin/bash
# Builds the library's host code with AddressSanitizer + UBSan and runs the host-side tests (mesh build, corner tables, Edgebreaker,
# sequencers, glTF container, connectivity decoder) plus a random-soup fuzz of dmi_mesh_build / dmi_encode_connectivity and a
# damaged-file fuzz of dmi_decode_connectivity against it.  CPU only.
set -e
cd "$(dirname "$0")/.."
make -s -C draco-oxide_amd/csrc
make -s -C draco-oxide_amd/csrc asan ASAN_DIR=/tmp/dmi_asan
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export DMI_LIBRARY=/tmp/dmi_asan/libdraco_mi_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1
python -m pytest tests/test_host_connectivity.py tests/test_gltf.py tests/test_decode_connectivity.py -x -q -m "not gpu"
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
import draco_oxide_amd as d
from draco_oxide_amd import synth
for n, ob in ((300, False), (257, True), (5, False)):
    d.encode_connectivity(synth.torus_mesh(n, open_boundary=ob))
rng = np.random.default_rng(1)
for trial in range(40):
    nv, nf = int(rng.integers(4, 3000)), int(rng.integers(1, 5000))
    pos = rng.integers(0, 6, size=(nv, 3)).astype(np.float32)
    if trial % 3 == 0:
        pos[rng.integers(0, nv)] = np.nan
    b = d.MeshBuilder()
    pid = b.add_attribute(pos, d.ATT_POSITION, d.DOMAIN_POSITION)
    b.add_attribute(rng.random((nv, 2)).astype(np.float32), d.ATT_TEXCOORD, d.DOMAIN_CORNER, parents=[pid])
    b.set_connectivity_attribute(rng.integers(0, nv, size=(nf, 3)).astype(np.uint32))
    try:
        d.encode_connectivity(b.build())
    except d.DracoMiError:
        pass   # an error code is fine; a sanitizer report is not
# the connectivity decoder on damaged files: truncations and random byte damage of files with topology splits, handles and seams
sys.path.insert(0, 'tests')
from test_decode_connectivity import _punched
for seed in range(6):
    conn = d.encode_connectivity(_punched(25, 0.1, seed, bool(seed & 1), True))
    good = conn.bytes
    conn.close()
    for cut in range(11, len(good), max(1, len(good) // 60)):
        try:
            d.decode_connectivity(good[:cut])
        except d.DracoMiError:
            pass
    for _ in range(1500):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 6))):
            b[int(rng.integers(11, len(b)))] = int(rng.integers(0, 256))
        try:
            d.decode_connectivity(bytes(b))
        except d.DracoMiError:
            pass
print("host sanitizers: clean")
PY
