import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import draco_oxide_amd as dmi
from draco_oxide_amd import synth
mesh = synth.torus_mesh(2236)
dm = dmi.DeviceMesh.upload(mesh, 0); cm = dm._c()
cfg = dmi.Config(device=0)
for _ in range(3):
    with dmi.encode_mesh_device_raw(dm, cfg, cm) as out: pass
for line in open("/proc/self/smaps_rollup"):
    if line.startswith(("Rss", "AnonHuge", "Anonymous")): print(line.strip())
print(open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "|", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
# mappings of ≥ 32 MB and how much of each is on huge pages
cur = None
rows = []
for line in open("/proc/self/smaps"):
    parts = line.split()
    if len(parts) >= 5 and "-" in parts[0] and parts[1][0] in "r-":
        cur = {"range": parts[0], "name": parts[5] if len(parts) > 5 else "", "Size": 0, "Rss": 0, "AnonHugePages": 0}
        rows.append(cur)
    elif cur is not None and parts and parts[0].rstrip(":") in ("Size", "Rss", "AnonHugePages"):
        cur[parts[0].rstrip(":")] = int(parts[1])
for r in rows:
    if r["Rss"] >= 32 * 1024:
        print(f'{r["range"]:>34} size {r["Size"] // 1024:6d} MB rss {r["Rss"] // 1024:6d} MB huge {r["AnonHugePages"] // 1024:6d} MB {r["name"]}')
