"""Writes tests/golden/drc/<fixture>.drc + tests/golden/manifest.json from the CPU oracle (source: "restatement") for the reference's
OBJ fixtures, encode::Config::default().  If a maintainer runs the Rust crate on the same fixtures (tests/compatibility.rs writes
tests/outputs/<name>.drc) and drops the files into tests/golden/reference_drc/, tests/test_golden_drc.py compares against THOSE —
the one-command upgrade from "parity unpinned" to pinned."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers

out_dir = os.path.join(ROOT, "tests", "golden", "drc")
os.makedirs(out_dir, exist_ok=True)
manifest = {}
for name in ("tetrahedron", "cube_quads", "sphere", "punctured_sphere", "torus"):
    blob = helpers.obj_session(name).encode(dump=False)
    open(os.path.join(out_dir, name + ".drc"), "wb").write(blob)
    manifest[name] = {"file": f"drc/{name}.drc", "bytes": len(blob), "sha256": hashlib.sha256(blob).hexdigest(), "source": "restatement",
                      "input": f"data/{name}.obj", "config": "encode::Config::default()"}
json.dump(manifest, open(os.path.join(ROOT, "tests", "golden", "manifest.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(manifest, indent=1))
