import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import draco_oxide_amd as dmi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2236
mesh = dmi.synth.torus_mesh(n)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    t = time.perf_counter(); conn = dmi.encode_connectivity(mesh); dt = time.perf_counter() - t
    print('encode_connectivity', round(dt, 3), 's', flush=True)
    conn.close()
