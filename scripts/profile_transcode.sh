#!/bin/bash
# Kernel trace of the transcode regime (1024 GLB files through dmi_transcode_assets, scripts/transcode_profile.py) → profiles/<tag>_transcode_kernel_stats.csv
set -u
tag=${1:-round}
calls=${2:-6}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$out/transcode_stats" -o st --output-format csv -- python3 "$root/scripts/transcode_profile.py" 1024 "$calls" > "$out/transcode_stats.log" 2>&1
tail -2 "$out/transcode_stats.log"
cd "$root"
python3 - "$out" "$tag" "$calls" <<'PY'
import csv, glob, re, sys
base, tag, calls = sys.argv[1], sys.argv[2], int(sys.argv[3])
hits = glob.glob(f"{base}/transcode_stats/**/*_kernel_stats.csv", recursive=True)
def short(n):
    m = re.search(r"(k_[a-z_0-9]+)", n)
    return m.group(1) if m else re.sub(r"\(.*", "", n)[:48]
rows = list(csv.DictReader(open(hits[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
out = [f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 scripts/transcode_profile.py 1024 {calls}   ({tag}, MI355X): {calls} transcodes of 1024 GLB files / 45.1M triangles (the first is the process's warm-up); kernel time summed over all streams {tot / calls:.1f} ms per call",
       "kernel, calls_per_transcode, ms_per_transcode, avg_us, min_us, pct"]
for r in rows:
    out.append(f"{short(r['Name'])}, {int(r['Calls']) / calls:.1f}, {float(r['TotalDurationNs']) / 1e6 / calls:.3f}, {float(r['AverageNs']) / 1e3:.2f}, {float(r['MinNs']) / 1e3:.2f}, {r['Percentage']}")
open(f"profiles/{tag}_transcode_kernel_stats.csv", "w").write("\n".join(out) + "\n")
print("\n".join(out[:45]))
PY
