#!/bin/bash
# Kernel durations of the 100M-triangle mesh (scripts/run_100m.py) for throwaway builds / env switches: VARIANTS="_a _b" ENVS="X=1" variant_100m.sh
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/variants_100m
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() {
  local tag=$1 lib=$2
  DMI_LIBRARY=$lib timeout 900 rocprofv3 --kernel-trace --stats -d "$out/$tag" -o s --output-format csv -- python3 "$root/scripts/run_100m.py" > "$out/$tag.log" 2>&1
  echo "== $tag"; grep -E "k_predict_packed|k_seq_quantize|k_tile_" "$out/$tag/s_kernel_stats.csv" | cut -d, -f1-4 | sed 's/dmi::(anonymous namespace):://'
}
run default "$root/draco-oxide_amd/libdraco_mi.so"
for v in ${VARIANTS:-}; do [ -f "$root/draco-oxide_amd/libdraco_mi$v.so" ] && run "v$v" "$root/draco-oxide_amd/libdraco_mi$v.so"; done
for e in ${ENVS:-}; do export $e; run "e_${e%%=*}" "$root/draco-oxide_amd/libdraco_mi.so"; unset ${e%%=*}; done
