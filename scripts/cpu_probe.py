import os, time, threading, ctypes, sys
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "n/a")
print("affinity:", len(os.sched_getaffinity(0)), "cpu_count:", os.cpu_count())
try:
    print("cpu.stat:", open("/sys/fs/cgroup/cpu.stat").read().replace("\n", " "))
except Exception as e:
    print(e)
# scaling of pure compute across threads (ctypes call releases the GIL): the library's host rANS coder on a fixed stream
sys.path.insert(0, ".")
import numpy as np, draco_oxide_amd as dmi
f = np.full(256, 4096 // 256, np.uint32); syms = np.random.default_rng(1).integers(0, 256, size=4_000_000).astype(np.uint32)
def work(): dmi.host_rans_stream(f, 12, syms)
for n in (1, 4, 8, 16, 32, 64, 128):
    th = [threading.Thread(target=work) for _ in range(n)]
    t = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; dt = time.perf_counter() - t
    print(f"{n:4d} threads: {dt:.3f} s  ({n * 4 / dt:.0f} Msym/s aggregate)")
