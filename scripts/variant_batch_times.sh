#!/bin/bash
# Kernel durations of the batch regime (scripts/bench_batch.py) for throwaway builds: VARIANTS="_a" variant_batch_times.sh [n_meshes]
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/variants_batch
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for v in "" ${VARIANTS:-}; do
  lib=$root/draco-oxide_amd/libdraco_mi$v.so
  [ -f "$lib" ] || continue
  DMI_LIBRARY=$lib rocprofv3 --kernel-trace --stats -d "$out/v$v" -o s --output-format csv -- python3 "$root/scripts/bench_batch.py" ${1:-1024} 5 > "$out/v$v.log" 2>&1
  echo "variant '$v'"; grep -v "^[EW]2026" "$out/v$v.log" | tail -1 | cut -c1-150
  grep -E "${KERNELS:-_multi}" "$out/v$v/s_kernel_stats.csv" | cut -d'"' -f2,3 | sed 's/dmi::(anonymous namespace):://; s/(.*)//' | cut -d, -f1-4 | head -12
done
