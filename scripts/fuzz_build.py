"""Fuzz of dmi_meshes_build (device MeshBuilder::build) against dmi_mesh_build (host builder, itself held to the oracle's restated builder by the
tests): batches of random primitives — rows drawn from small pools (duplicates), ±0.0, NaN rows, points copied whole (byte-identical NaN rows), NaN payloads, constant attributes, strided rows, u8/u16/u32
indices, degenerate faces, unreferenced points, 1–5 attributes, Position not first.  usage: fuzz_build.py [batches] [first_seed]; tests/test_gpu_fuzz_slice.py
runs a seeded slice under -m gpu."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import draco_oxide_amd as dmi  # noqa: E402


def random_primitive(rng):
    n_pts = int(rng.integers(3, 250)) if rng.random() < 0.5 else int(rng.integers(250, 6000))
    n_faces = int(rng.integers(1, 3 * n_pts))
    pool_n = max(2, int(n_pts * rng.choice([0.05, 0.3, 0.9, 3.0])))
    kinds = [dmi.ATT_POSITION]
    for t in (dmi.ATT_NORMAL, dmi.ATT_TEXCOORD, dmi.ATT_CUSTOM, dmi.ATT_COLOR):
        if rng.random() < 0.5:
            kinds.append(t)
    rng.shuffle(kinds)
    pos_at = kinds.index(dmi.ATT_POSITION)
    specs = []
    for t in kinds:
        ncomp = {dmi.ATT_POSITION: 3, dmi.ATT_NORMAL: 3, dmi.ATT_TEXCOORD: 2, dmi.ATT_CUSTOM: 1, dmi.ATT_COLOR: int(rng.integers(1, 5))}[t]
        if t == dmi.ATT_CUSTOM:
            rows = rng.integers(0, max(2, pool_n // 4), size=(n_pts, 1)).astype(np.uint32)
        else:
            pool = rng.integers(-3, 4, size=(pool_n, ncomp)).astype(np.float32) * np.float32(0.5)
            rows = pool[rng.integers(0, pool_n, size=n_pts)].copy()
            if rng.random() < 0.3:
                rows[rng.integers(0, n_pts, size=max(1, n_pts // 20))] *= np.float32(-0.0)       # -0.0 / 0.0 rows
            if rng.random() < 0.3:
                rows[rng.integers(0, n_pts, size=max(1, n_pts // 30)), int(rng.integers(0, ncomp))] = np.nan
            if rng.random() < 0.1:
                rows[:] = rows[0]                                                                  # one value for every point
        if rng.random() < 0.3 and rows.dtype == np.float32:                                      # an interleaved buffer: strided rows
            wide = np.zeros((n_pts, ncomp + int(rng.integers(1, 4))), np.float32)
            wide[:, :ncomp] = rows
            rows = wide[:, :ncomp]
        dom = dmi.DOMAIN_POSITION if t == dmi.ATT_POSITION else dmi.DOMAIN_CORNER
        par = [pos_at] if t in (dmi.ATT_NORMAL, dmi.ATT_TEXCOORD) else []
        specs.append((rows, t, dom, par))
    if rng.random() < 0.5:   # points that are copies of other points in EVERY attribute — NaN rows included: byte-identical NaN rows merge (builder.rs:254-279)
        k = max(1, n_pts // 8)
        src, dst = rng.integers(0, n_pts, size=k), rng.integers(0, n_pts, size=k)
        for rows, *_ in specs:
            for s_, d_ in zip(src, dst):
                rows[d_] = rows[s_]
        if rng.random() < 0.5:   # … some of them with another NaN payload or the other zero: different bytes, no merge for a NaN row
            for rows, *_ in specs:
                if rows.dtype == np.float32 and rng.random() < 0.5:
                    for d_ in dst[: max(1, k // 2)]:
                        row = rows[d_]
                        bits = np.ascontiguousarray(row).view(np.uint32).copy()
                        bits[np.isnan(row)] ^= np.uint32(1)
                        bits[row == 0] ^= np.uint32(0x80000000)
                        rows[d_] = bits.view(np.float32)
    hi = n_pts if rng.random() < 0.7 else max(3, n_pts - int(rng.integers(1, 10)))   # sometimes the last points stay unreferenced
    faces = rng.integers(0, hi, size=(n_faces, 3)).astype(np.uint32)
    if rng.random() < 0.5:
        faces[rng.integers(0, n_faces), 1] = faces[rng.integers(0, n_faces), 0]
    dt = np.uint8 if hi <= 255 and rng.random() < 0.5 else (np.uint16 if hi <= 65535 and rng.random() < 0.6 else np.uint32)
    return specs, faces, dt


def same(a, b):
    if a.faces.shape != b.faces.shape or not (a.faces == b.faces).all() or len(a.attributes) != len(b.attributes):
        return False
    for x, y in zip(a.attributes, b.attributes):
        if x.values.shape != y.values.shape or x.values.tobytes() != y.values.tobytes() or (x.point_to_value is None) != (y.point_to_value is None):
            return False
        if x.point_to_value is not None and not (x.point_to_value == y.point_to_value).all():
            return False
        if (x.num_points, x.unique_id, x.att_type, x.domain, x.parent_index) != (y.num_points, y.unique_id, y.att_type, y.domain, y.parent_index):
            return False
    return True


def run(n_batches, seed0, log=print):
    """→ (mismatches, primitives, built on the device, built by the host builder inside the batch call)"""
    bad = prims = n_dev = n_host = 0
    for b in range(n_batches):
        rng = np.random.default_rng(seed0 + b)
        raws, builders = [], []
        for _ in range(int(rng.integers(1, 24))):
            specs, faces, dt = random_primitive(rng)
            rm, mb = dmi.RawMesh(), dmi.MeshBuilder()
            for rows, t, d, par in specs:
                rm.add_attribute(rows, t, d, par)
                mb.add_attribute(rows, t, d, parents=par)
            rm.set_indices(faces.astype(dt).ravel())
            mb.set_connectivity_attribute(faces)
            raws.append(rm)
            builders.append(mb)
        with dmi.meshes_build(raws, host_values=True) as batch:
            tm = dmi.last_build_timings()
            n_dev += tm["device_meshes"]; n_host += tm["host_meshes"]
            for j, mb in enumerate(builders):
                prims += 1
                if not same(batch.mesh(j), mb.build()):
                    log(f"batch {b} (seed {seed0 + b}) primitive {j}: the device-built mesh differs from the host builder's"); bad += 1
    return bad, prims, n_dev, n_host


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    bad, prims, n_dev, n_host = run(n, s0)
    print(f"{n} batches, {prims} primitives ({n_dev} built by the kernels, {n_host} by the host builder inside the call), {bad} mismatches")
    sys.exit(1 if bad else 0)
