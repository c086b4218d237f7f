#!/bin/bash
# Quick GPU check of a kernel change (gpurun from the repo root): scripts/r6_quick.sh <tag> [pytest args...]
# the tests named (default: the whole-mesh + parity files), then a kernel trace of 5 bench steps → gpurun_out/<tag>/{tests.log,stats.csv,bench.log}
set -u
tag=${1:-q}; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
tests=${*:-tests/test_gpu_device_mesh.py tests/test_gpu_parity.py tests/test_gpu_configs.py}
timeout 1500 python3 -m pytest $tests -x -q -m gpu > "$out/tests.log" 2>&1
echo "tests rc=$?"; tail -3 "$out/tests.log"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$out/stats" -o st --output-format csv -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-batch --no-scopes --no-traffic > "$out/bench.log" 2>&1
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, re, json
out = sys.argv[1]
hits = glob.glob(out + "/stats/**/*_kernel_stats.csv", recursive=True)
if hits:
    rows = list(csv.DictReader(open(hits[0])))
    with open(out + "/stats.csv", "w") as f:
        for r in rows[:40]:
            m = re.search(r"(k_[a-z_0-9]+)", r["Name"]); n = m.group(1) if m else r["Name"][:40]
            line = f"{n:32s} calls {int(r['Calls']):4d} avg_us {float(r['AverageNs'])/1e3:9.2f} min_us {float(r['MinNs'])/1e3:9.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.3f}"
            print(line); f.write(line + "\n")
for l in open(out + "/bench.log"):
    if l.startswith("{"):
        b = json.loads(l); print(json.dumps({k: b[k] for k in ("value", "ms_per_step", "roofline", "stages_ms")}))
PY
