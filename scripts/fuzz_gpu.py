"""One-off GPU fuzz (needs an MI355X): random triangle soups with normals and UVs, heavy-tailed grids and tiny meshes at random
quantization widths, each encoded through the whole-mesh call, the device-table batch call and the host-table form, compared with the
oracle (test infrastructure).  usage: fuzz_gpu.py [cases] [first_seed]; tests/test_gpu_fuzz_slice.py runs a seeded slice under -m gpu."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401  (one HIP runtime for torch and the library)
import draco_oxide_amd as dmi
import orc
from helpers import oracle_from_product_mesh, oracle_values_by_point
import test_gpu_parity as T
from test_gpu_decode import numpy_quantize
from test_gpu_decode_mesh import _canonical_faces, _requantize

def run(n_cases, seed0, log=print):
  """n_cases seeded cases → (mismatches, rejected by the reference algorithm, decoded by both decoders, whole files read back)."""
  rng = np.random.default_rng(seed0)
  bad = rejected = decoded = whole = 0
  for c in range(n_cases):
      seed = seed0 + c
      kind = c % 3
      if kind == 0:
          mesh, sess = T._soup_mesh(seed, uv_per_corner=bool(c & 1))
      elif kind == 1:
          mesh = T._heavy_tailed_mesh(int(rng.integers(3, 60)), seed)
          sess = oracle_from_product_mesh(mesh)
      else:
          mesh = dmi.synth.torus_mesh(int(rng.integers(3, 40)), seed=seed, open_boundary=bool(c & 2), normals=bool(c & 4), uvs=bool(c & 8) or not (c & 4))
          sess = oracle_from_product_mesh(mesh)
      pb, ub = int(rng.integers(1, 21)), int(rng.integers(1, 17))
      cfg = dmi.Config(pos_bits=pb, uv_bits=ub)
      try:
          want = sess.encode(pos_bits=pb, uv_bits=ub)
      except orc.OracleError:
          rejected += 1
          try:
              dmi.encode_mesh(mesh, cfg)
              log(f"case {c} (seed {seed}, {pb}/{ub} bits): the oracle rejects, the library does not"); bad += 1
          except dmi.DracoMiError:
              pass
          continue
      got = {"whole": dmi.encode_mesh(mesh, cfg)}
      jobs = dmi.meshes_prepare([mesh, mesh], cfg)
      outs = dmi.jobs_encode(jobs)
      got["batch"] = jobs[1].header_and_connectivity + outs[1]
      os.environ["DMI_HOST_TABLES"] = "1"
      got["host tables"] = dmi.encode_mesh(mesh, cfg)
      del os.environ["DMI_HOST_TABLES"]
      # round 3: the mesh resident in HBM (device corner tables whatever the size; flagged meshes fall back to the reference's walks) and the
      # per-mesh host form of the batch prepare
      got["mesh in HBM"] = dmi.encode_mesh_device(dmi.DeviceMesh.upload(mesh), cfg)
      # round 4: MeshBuilder::build on the device from per-point rows, then the connectivity stage on the resident result
      rm = dmi.RawMesh()
      for a in mesh.attributes:
          rm.add_attribute(a.values if a.point_to_value is None else a.values[a.point_to_value], a.att_type, a.domain, [] if a.parent_index < 0 else [a.parent_index])
      rm.set_indices(mesh.faces.ravel())
      with dmi.meshes_build([rm, rm], cfg) as built:
          if built.num_faces(1) == len(mesh.faces):   # (rows that repeat make the builder merge points: then it is another mesh than the oracle's session holds)
              jb = dmi.built_meshes_prepare(built, [1], cfg)
              got["device build"] = jb[0].header_and_connectivity + dmi.jobs_encode(jb)[0]
              jb[0].close()
      os.environ["DMI_HOST_CONNECTIVITY"] = "1"
      jobs2 = dmi.meshes_prepare([mesh], cfg)
      got["batch, host connectivity"] = jobs2[0].header_and_connectivity + dmi.jobs_encode(jobs2)[0]
      del os.environ["DMI_HOST_CONNECTIVITY"]
      for j in jobs2:
          j.close()
      for j in jobs:
          j.close()
      for name, g in got.items():
          if g != want:
              log(f"case {c} (seed {seed}, kind {kind}, {pb}/{ub} bits, {len(mesh.faces)} faces): {name} differs ({len(g)} vs {len(want)} bytes)"); bad += 1
      # the product's decoder against the oracle's, value for value (by point)
      try:
          section = outs[1]
          conn = dmi.encode_connectivity(mesh)
          tables = [conn.table(i) for i in range(conn.num_tables)]
          dec = dmi.decode_attributes(section, tables, mesh.attributes[0].num_points, seeds=conn.seeds())
          conn.close()
          ref, used = sess.decode_attributes(section)
          assert used == len(section) and len(ref) == len(dec)
          for i, (g, d) in enumerate(zip(dec, ref)):
              per_point, seen = oracle_values_by_point(tables[i], tables[0], d, len(g["values"]))
              if g["portabilization"] == 3:
                  ok = np.abs(g["values"][seen] - per_point[seen]).max(initial=0) < 2e-6
              else:
                  ok = (g["values"][seen].view(np.uint32) == per_point[seen].view(np.uint32)).all()
              if not ok:
                  log(f"case {c} (seed {seed}): decoded attribute {i} differs from the oracle decoder"); bad += 1
          decoded += 1
      except (dmi.DracoMiError, orc.OracleError, AssertionError) as e:
          log(f"case {c} (seed {seed}, kind {kind}, {pb}/{ub} bits): decoder: {type(e).__name__} {str(e)[:120]}"); bad += 1
      # the whole file from its bytes alone: the decoded triangles are the input's (quantized position rows, labelling-free)
      try:
          dm = dmi.decode_mesh(want)
          pos = mesh.attributes[0]
          q, mn, rg = numpy_quantize(pos.values, pb)
          q = q if pos.point_to_value is None else q[pos.point_to_value]
          in_faces = np.asarray(mesh.faces, np.int64).reshape(-1, 3)
          got_rows = _requantize(dm["attributes"][0]["values"], mn, rg, pb)[dm["faces"].astype(np.int64)]
          assert dm["faces"].shape == in_faces.shape and (_canonical_faces(q[in_faces]) == _canonical_faces(got_rows)).all()
          whole += 1
      except (dmi.DracoMiError, AssertionError) as e:
          log(f"case {c} (seed {seed}, kind {kind}, {pb}/{ub} bits): decode_mesh: {type(e).__name__} {str(e)[:120]}"); bad += 1
  return bad, rejected, decoded, whole


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    bad, rejected, decoded, whole = run(n_cases, seed0)
    print(f"{whole} whole files read back by dmi_decode_mesh")
    print(f"{n_cases} cases, {rejected} rejected by the reference algorithm, {decoded} decoded back by both decoders, {bad} mismatches")
    sys.exit(1 if bad else 0)
