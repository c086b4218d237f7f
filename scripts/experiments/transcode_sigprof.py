"""round 6: where the CPU time of the 1024-file transcode goes (sampling profiler: sigprof.c).  python3 transcode_sigprof.py [plain|seams] [calls=8] [hz=2000]"""
import os, sys, time, collections, ctypes as C, subprocess, re
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import draco_oxide_amd as dmi
from draco_oxide_amd import synth, gltf, binding
kind = sys.argv[1] if len(sys.argv) > 1 else "plain"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hz = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
sp = C.CDLL(os.path.join(HERE, "libsigprof.so"))
binding.configure_process(huge_page_new=True, numa_pin=True)
cfg = dmi.Config(device=0)
if kind == "batch":      # bench.py's batch_regime: 256 host meshes -> dmi_meshes_prepare + dmi_jobs_encode
    meshes = synth.batch_meshes(256)
    total = sum(len(m.faces) for m in meshes)
    def one():
        jobs = dmi.meshes_prepare(meshes, cfg)
        with dmi.jobs_encode_raw(jobs):
            pass
        for j in jobs:
            j.close()
elif kind == "decode":   # dmi_decode_mesh of a 3M-triangle file (the validation-side decoder)
    mesh = synth.torus_mesh(1225)
    total = len(mesh.faces)
    drc = dmi.encode_mesh(mesh, cfg)
    def one():
        dmi.decode_mesh(drc)
elif kind == "single":   # bench.py's value: one 10M-triangle mesh in HBM -> whole .drc
    mesh = synth.torus_mesh(2236)
    total = len(mesh.faces)
    dm = dmi.DeviceMesh.upload(mesh)
    def one():
        dmi.encode_mesh_device(dm, cfg)
else:
    glbs, total = synth.batch_glbs(1024, seams=(kind == "seams"))
    alist = binding.AssetList(glbs)
    def one():
        gltf.transcode_files(alist, cfg)
for _ in range(3):
    one()
sp.sp_start(hz)
t0 = time.perf_counter()
for _ in range(calls):
    one()
wall = time.perf_counter() - t0
n = sp.sp_stop()
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", f"sigprof_{kind}.txt")
os.makedirs(os.path.dirname(out), exist_ok=True)
sp.sp_dump(out.encode(), 10)
print(f"{kind}: {calls} calls, {wall / calls * 1e3:.1f} ms per call, {n} samples at {hz} Hz = {n / hz / calls * 1e3:.0f} ms of CPU per call")
lines = [l.rstrip("\n") for l in open(out)]
# internal functions of libdraco_mi.so are not dynamic symbols (version script): name its frames from the file's symbol table with llvm-symbolizer
lib = binding.library_path()
offs = sorted({m.group(1) for l in lines for m in re.finditer(r"libdraco_mi\.so![^+ ]*\+(0x[0-9a-f]+)", l)})
names = {}
if offs:
    sym = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"
    res = subprocess.run([sym, "--obj=" + lib, "--functions=linkage", "--demangle", "--no-inlines", "--output-style=LLVM"] + offs, capture_output=True, text=True).stdout.split("\n\n")
    for o, r in zip(offs, res):
        rl = r.strip().split("\n") if r.strip() else ["?"]
        fn = rl[0]
        if os.environ.get("SIGPROF_LINES") and len(rl) > 1 and not rl[1].startswith("??"):
            fn = fn.split("(")[0] + "@" + os.path.basename(rl[1])
        names[o] = re.sub(r"\(.*", "", fn.replace("(anonymous namespace)::", "").replace("'lambda", "{lambda"))[:110]
def nice(fr):
    m = re.match(r"libdraco_mi\.so![^+ ]*\+(0x[0-9a-f]+)", fr)
    return "libdraco_mi.so!" + names.get(m.group(1), "?") if m else fr.split("+0x")[0]
lines = [" < ".join(nice(fr) for fr in l.split(" < ")) for l in lines]
top = collections.Counter(l.split(" < ")[0] for l in lines)
print("---- self (innermost frame)")
for k, v in top.most_common(45):
    print(f"{v / n * 100:6.2f} %  {v / hz / calls * 1e3:7.1f} ms/call  {k[:150]}")
def first_ours(l):
    for fr in l.split(" < "):
        if fr.startswith("libdraco_mi.so!") and not fr.endswith("!?"):
            return fr
    return "(none of libdraco_mi.so in the top frames) " + l.split(" < ")[0]
incl = collections.Counter(first_ours(l) for l in lines)
print("---- attributed to the first named libdraco_mi.so frame up the stack")
for k, v in incl.most_common(45):
    print(f"{v / n * 100:6.2f} %  {v / hz / calls * 1e3:7.1f} ms/call  {k[:150]}")

rt = [l for l in lines if l.split("!")[0] in ("libhsa-runtime64.so", "libamdhip64.so", "libhsakmt.so") or l.startswith("libc.so.6!ioctl")]
inc2 = collections.Counter(first_ours(l) for l in rt)
print(f"---- samples inside the HIP / HSA runtime or its ioctls ({len(rt)} = {len(rt) / n * 100:.1f} %): the libdraco_mi.so function that called in")
for k, v in inc2.most_common(30):
    print(f"{v / n * 100:6.2f} %  {v / hz / calls * 1e3:7.1f} ms/call  {k[:150]}")
libc = [l for l in lines if l.startswith("libc.so.6!?")]
inc3 = collections.Counter(first_ours(l) for l in libc)
print(f"---- samples in unnamed libc code (memcpy / memset ...: {len(libc)} = {len(libc) / n * 100:.1f} %): the libdraco_mi.so function that called in")
for k, v in inc3.most_common(20):
    print(f"{v / n * 100:6.2f} %  {v / hz / calls * 1e3:7.1f} ms/call  {k[:150]}")

def chain(l, k=3):
    out = [fr for fr in l.split(" < ") if fr.startswith("libdraco_mi.so!")][:k]
    return " < ".join(f.replace("libdraco_mi.so!", "") for f in out) or "(no libdraco_mi.so frame)"
inc4 = collections.Counter(chain(l) for l in libc)
print("---- the same samples by their three innermost libdraco_mi.so frames")
for k, v in inc4.most_common(25):
    print(f"{v / n * 100:6.2f} %  {v / hz / calls * 1e3:7.1f} ms/call  {k[:230]}")
