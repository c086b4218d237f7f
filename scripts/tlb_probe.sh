#!/bin/bash
# Address-translation counters of the pass kernels at 10M and 100M triangles (the 100M-triangle mesh runs the quantize+predict pass at
# 23.5 % of the roofline against 30 % at 10M): TCP_UTCL1 requests / hits / misses per launch.  gpurun -- bash scripts/tlb_probe.sh
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r3/tlb
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for n in 2236 7071; do
  rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_GUI_ACTIVE --kernel-include-regex "k_predict_packed|k_seq_quantize|k_value_ranges|k_histogram" -d "$out/n$n" -o t --output-format csv -- python3 "$root/scripts/resident_steps.py" $n 3 > "$out/n$n.log" 2>&1
done
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, re, sys, collections
base = sys.argv[1]
for n in (2236, 7071):
    hits = glob.glob(f"{base}/n{n}/**/*_counter_collection.csv", recursive=True)
    if not hits:
        print(n, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(hits[0])):
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        acc[m.group(1) if m else r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== n = {n} ({2 * n * n} triangles)")
    for k, c in sorted(acc.items()):
        avg = {name: sum(v) / len(v) for name, v in c.items()}
        req, hit, miss = avg.get("TCP_UTCL1_REQUEST_sum", 0), avg.get("TCP_UTCL1_TRANSLATION_HIT_sum", 0), avg.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0)
        print(f"{k}: utcl1 requests {req:.0f}, hits {hit:.0f}, misses {miss:.0f} ({100 * miss / max(req, 1):.2f} % of requests), busy cycles {avg.get('GRBM_GUI_ACTIVE', 0):.0f}")
PY
