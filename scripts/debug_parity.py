import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import draco_oxide_amd as dmi
from draco_oxide_amd import synth
import helpers, orc

for n in [int(x) for x in sys.argv[1:]]:
    mesh = synth.torus_mesh(n)
    t0 = time.time(); job = dmi.mesh_prepare(mesh, dmi.Config(flags=dmi.FLAG_TIMINGS)); t1 = time.time()
    got = job.header_and_connectivity + job.encode(); t2 = time.time()
    tm = job.timings()
    sess = helpers.oracle_from_product_mesh(mesh); t3 = time.time()
    want = sess.encode(); t4 = time.time()
    same = got == want
    print(f"n={n} F={len(mesh.faces)} prepare={t1-t0:.2f}s encode={t2-t1:.3f}s oracle_build={t3-t2:.2f}s oracle_encode={t4-t3:.2f}s same={same} len={len(got)}/{len(want)}", tm, flush=True)
    if not same:
        k = next(i for i in range(min(len(got), len(want))) if got[i] != want[i])
        print("first diff at", k, "conn len", len(job.header_and_connectivity))
        for i in range(3):
            blk = bytes(sess.blob(f"att{i}.bytes"))
            pos = want.find(blk)
            print("att", i, "block at", pos, "len", len(blk))
