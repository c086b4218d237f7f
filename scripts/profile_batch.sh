#!/bin/bash
# Kernel trace of the batch regime (256 meshes: dmi_meshes_prepare + dmi_jobs_encode, scripts/bench_batch.py) → gpurun_out/<tag>/batch_stats
set -u
tag=${1:-round}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$out/batch_stats" -o st --output-format csv -- python3 "$root/scripts/bench_batch.py" 256 5 > "$out/batch_stats.log" 2>&1
tail -2 "$out/batch_stats.log"
cd "$root"
python3 - "$out" "$tag" <<'PY'
import csv, glob, re, sys
base, tag = sys.argv[1], sys.argv[2]
hits = glob.glob(f"{base}/batch_stats/**/*_kernel_stats.csv", recursive=True)
def short(n):
    m = re.search(r"(k_[a-z_0-9]+)", n)
    return m.group(1) if m else re.sub(r"\(.*", "", n)[:48]
out = [f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 scripts/bench_batch.py 256 5   ({tag}, MI355X): 1 dmi_meshes_prepare + 12 dmi_jobs_encode of 256 meshes (11.1M triangles)", "kernel, calls, total_ms, avg_us, pct"]
for r in csv.DictReader(open(hits[0])):
    out.append(f"{short(r['Name'])}, {r['Calls']}, {float(r['TotalDurationNs']) / 1e6:.3f}, {float(r['AverageNs']) / 1e3:.2f}, {r['Percentage']}")
open(f"profiles/{tag}_batch_kernel_stats.csv", "w").write("\n".join(out) + "\n")
print("\n".join(out[:40]))
PY
