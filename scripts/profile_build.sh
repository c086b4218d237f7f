#!/bin/bash
# Kernel trace of the device mesh build (scripts/build_time.py) on the GPU box: scripts/profile_build.sh <tag> [grid] [n_batch]
set -u
tag=${1:-build}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$out/stats" -o st --output-format csv -- python3 "$root/scripts/build_time.py" ${2:-2236} ${3:-256} > "$out/stats.log" 2>&1
cd "$root"
f=$(ls $out/stats/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -z "$f" ] && f=$(ls $out/stats/*kernel_stats.csv 2>/dev/null | head -1)
echo "stats file: $f"
head -40 "$f" | cut -c1-200
tail -12 "$out/stats.log"
