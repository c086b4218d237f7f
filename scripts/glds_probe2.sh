#!/bin/bash
# DMI_SWEEP_GLDS + 8 waves per SIMD against the default build: the three sweep instantiations at 10M triangles, and the batch regime
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/glds2
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for v in "" ${VARIANTS:-_glds8}; do
  lib=$root/draco-oxide_amd/libdraco_mi$v.so
  [ -f "$lib" ] || continue
  DMI_LIBRARY=$lib rocprofv3 --kernel-trace --stats -d "$out/v$v" -o s --output-format csv -- python3 "$root/scripts/sweep_ablation.py" 2236 3 > "$out/v$v.log" 2>&1
  echo "variant '$v'"; grep -E "k_predict_packed" "$out/v$v/s_kernel_stats.csv" | cut -d, -f1-5
  DMI_LIBRARY=$lib rocprofv3 --kernel-trace --stats -d "$out/b$v" -o s --output-format csv -- python3 "$root/scripts/bench_batch.py" 256 5 > "$out/b$v.log" 2>&1
  tail -3 "$out/b$v.log"; grep -E "k_predict_packed" "$out/b$v/s_kernel_stats.csv" | cut -d, -f1-5
done
