"""Where does dmi_meshes_prepare spend its time?  Aggregates the DMI_TRACE lines of one 1024-mesh call."""
import os, re, subprocess, sys, collections
code = ("import sys, time; sys.path.insert(0, %r); import draco_oxide_amd as dmi; from draco_oxide_amd import synth; meshes = synth.batch_meshes(1024)\n"
        "w = dmi.meshes_prepare(meshes[:16]); [j.close() for j in w]\n"
        "print('MARK', flush=True); sys.stderr.write('MARK\\n'); sys.stderr.flush()\n"
        "t = time.perf_counter(); jobs = dmi.meshes_prepare(meshes); print('prepare_s', round(time.perf_counter() - t, 3))\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DMI_TRACE="1", DMI_TRACE_TABLES="1"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
out = out[out.rindex("MARK"):]
acc = collections.defaultdict(float); n = 0
for l in out.splitlines():
    m = re.search(r"host connectivity of (\d+) faces.*universal corner table ([\d.]+) ms, attribute tables ([\d.]+), Edgebreaker ([\d.]+), universal sequencer ([\d.]+), seam-table sequencers \+ views ([\d.]+); total ([\d.]+)", l)
    if m:
        for k, v in zip(("conn_universal", "conn_att_tables", "conn_edgebreaker", "conn_sequencer", "conn_views", "conn_total"), m.groups()[1:]): acc[k] += float(v)
        n += 1
    m = re.search(r"universal table of \d+ faces: copy \+ vertex ids ([\d.]+) ms, half-edge matching ([\d.]+) \((.*?)\), left-most corners ([\d.]+)", l)
    if m:
        acc["  universal: copy + vertex ids"] += float(m.group(1)); acc["  universal: half-edge matching"] += float(m.group(2)); acc["  universal: left-most corners"] += float(m.group(4))
    m = re.search(r"job create \((\d+) faces, (\w+) relabelling\): sequences ([\d.]+) ms, relabel \+ table uploads ([\d.]+), attribute uploads \+ buffers \+ fan rows ([\d.]+), stream \+ plan ([\d.]+)", l)
    if m:
        for k, v in zip(("create_validate", "create_relabel_uploads", "create_attr_buffers", "create_stream_plan"), m.groups()[2:]): acc[k] += float(v)
print([l for l in out.splitlines() if l.startswith("prepare_s")])
print("meshes", n, "thread-milliseconds summed over all meshes:")
for k, v in acc.items(): print(f"  {k:26s} {v:10.1f} ms")
