"""Host connectivity stage of one large mesh, stage by stage (DMI_TRACE lines of libdraco_mi) — default form and DMI_SERIAL_TABLES=1.
usage: python scripts/prep_breakdown.py [grid n]   (no GPU needed)"""
import os, subprocess, sys
n = sys.argv[1] if len(sys.argv) > 1 else "2236"
code = ("import sys, time; sys.path.insert(0, %r); import draco_oxide_amd as dmi; from draco_oxide_amd import synth; m = synth.torus_mesh(%s)\n"
        "for _ in range(3):\n    t = time.time(); c = dmi.encode_connectivity(m); print('dmi_encode_connectivity', round(time.time() - t, 3), 's', flush=True); c.close()\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), n)
for label, extra in (("default", {}), ("DMI_SERIAL_TABLES=1", {"DMI_SERIAL_TABLES": "1"})):
    print("==", label)
    env = dict(os.environ, DMI_TRACE="1", **extra)
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    print("\n".join(l for l in out.splitlines() if "host connectivity" in l or "dmi_encode_connectivity" in l))
