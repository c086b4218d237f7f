#!/bin/bash
# A/B of in-tree library builds on the bench workload: scripts/r6_ab.sh <tag> <suffix...>  ("base" = libdraco_mi.so)
# per variant: a kernel trace of 5 bench steps → the pass's kernels (avg / min µs) and the bench line's roofline numbers
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$root/draco-oxide_amd/libdraco_mi.so; [ "$v" != base ] && lib=$lib.$v
  DMI_LIBRARY=$lib timeout 600 rocprofv3 --kernel-trace --stats -d "$out/$v" -o st --output-format csv -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-batch --no-scopes --no-traffic > "$out/$v.log" 2>&1
  python3 - "$out" "$v" <<'PY'
import csv, glob, sys, re, json
out, v = sys.argv[1], sys.argv[2]
hits = glob.glob(f"{out}/{v}/**/*_kernel_stats.csv", recursive=True)
keep = ("k_value_ranges", "k_value_quantize_rec", "k_seq_gather_rec", "k_predict_packed", "k_texcoord_fixup", "k_build_fans", "k_histogram")
line = [v]
if hits:
    for r in csv.DictReader(open(hits[0])):
        m = re.search(r"(k_[a-z_0-9]+)", r["Name"])
        if m and m.group(1).startswith(keep):
            line.append(f"{m.group(1)} {float(r['AverageNs'])/1e3:.1f}/{float(r['MinNs'])/1e3:.1f}")
for l in open(f"{out}/{v}.log"):
    if l.startswith("{"):
        b = json.loads(l); r = b["roofline"]
        line.append(f"| value {b['value']} frac {r['frac']} early {r['early_stage_ms']} after {r['after_walks_ms']}")
print("  ".join(line))
open(f"{out}/summary.txt", "a").write("  ".join(line) + "\n")
PY
done
