"""Aggregates the DMI_TRACE lines of dmi_meshes_prepare (host connectivity + job creation) from a log on stdin."""
import re, sys
import numpy as np
conn, jc = [], []
for l in sys.stdin:
    m = re.search(r'host connectivity of (\d+) faces: universal corner table ([\d.]+) ms, attribute tables ([\d.]+), Edgebreaker ([\d.]+), sequencers ([\d.]+)', l)
    if m: conn.append([float(x) for x in m.groups()])
    m = re.search(r'job create \((\d+) faces\): sequences ([\d.]+) ms, relabel \+ table uploads ([\d.]+), attribute uploads \+ buffers \+ fan rows ([\d.]+), stream \+ plan ([\d.]+)', l)
    if m: jc.append([float(x) for x in m.groups()])
c, j = np.array(conn), np.array(jc)
print(f"meshes {len(c)} faces {c[:,0].sum():.0f}: thread-time sums (ms): connectivity {c[:,1:].sum():.0f} (tables {c[:,1].sum()+c[:,2].sum():.0f}, Edgebreaker {c[:,3].sum():.0f}, sequencers {c[:,4].sum():.0f}); "
      f"job create {j[:,1:].sum():.0f} (relabel + table uploads {j[:,2].sum():.0f}, attribute uploads + buffers + fan rows {j[:,3].sum():.0f}, stream + plan {j[:,4].sum():.0f})")
