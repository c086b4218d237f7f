"""A seeded slice of scripts/fuzz_gpu.py under -m gpu (the fuzz evidence must be driver-run, not hand-run): random triangle
soups, heavy-tailed grids and tiny meshes at random quantization widths, each through six encode forms (whole mesh, batch, host tables,
mesh resident in HBM, device-built mesh, batch with host connectivity) against the oracle, both attribute decoders, and dmi_decode_mesh."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_gpu", os.path.join(ROOT, "scripts", "fuzz_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed0", [41000, 42000])
def test_fuzz_slice(seed0):
    lines = []
    bad, rejected, decoded, whole = _fuzz().run(250, seed0, log=lines.append)
    assert bad == 0, "\n".join(lines[:20])
    assert decoded + rejected == 250 and whole == decoded


def test_fuzz_slice_of_the_device_mesh_build():
    """scripts/fuzz_build.py: 40 seeded batches of random primitives through dmi_meshes_build against the host builder."""
    spec = importlib.util.spec_from_file_location("fuzz_build", os.path.join(ROOT, "scripts", "fuzz_build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = []
    bad, prims, n_dev, n_host = mod.run(40, 88000, log=lines.append)
    assert bad == 0, "\n".join(lines[:20])
    assert prims > 200 and n_dev > prims // 2
