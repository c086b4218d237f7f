#!/bin/bash
# SQ / TCP counters of the sweep for the resident 10M-triangle workload, default build vs an env switch (e.g. DMI_NO_P48=1)
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/sq_probe
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for v in "" "$@"; do
  i=$((i+1))
  [ -n "$v" ] && export $v
  timeout 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU -d "$out/sq$i" -o s --output-format csv -- python3 "$root/scripts/resident_steps.py" 2236 2 > "$out/sq$i.log" 2>&1
  timeout 240 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE -d "$out/tcp$i" -o t --output-format csv -- python3 "$root/scripts/resident_steps.py" 2236 2 > "$out/tcp$i.log" 2>&1
  [ -n "$v" ] && unset ${v%%=*}
  echo "== '${v:-default}'"
  python3 - "$out" $i <<'PY'
import csv, glob, sys, collections
base, i = sys.argv[1], sys.argv[2]
for kind in ("sq", "tcp"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob(f"{base}/{kind}{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_predict_packed" not in k and "k_seq_quantize" not in k: continue
            import re
            k = re.search(r"(k_[a-z_0-9]+)", k).group(1)
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            if r["Counter_Name"] in ("SQ_WAVES", "TCP_TOTAL_CACHE_ACCESSES_sum"): cnt[k] += 1
    for k, c in acc.items():
        n = max(cnt[k], 1)
        print(kind, k, {a: round(b / n) for a, b in c.items()})
PY
done
