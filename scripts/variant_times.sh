#!/bin/bash
# Kernel durations of the 10M-triangle resident workload for throwaway builds (scripts/build_variant.sh): VARIANTS="_a _b" variant_times.sh
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/variants
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for v in "" ${VARIANTS:-}; do
  lib=$root/draco-oxide_amd/libdraco_mi$v.so
  [ -f "$lib" ] || continue
  DMI_LIBRARY=$lib rocprofv3 --kernel-trace --stats -d "$out/v$v" -o s --output-format csv -- python3 "$root/scripts/sweep_ablation.py" 2236 ${SUBSETS:-1} > "$out/v$v.log" 2>&1
  echo "variant '$v'"; grep -E "${KERNELS:-k_predict_packed|k_seq_quantize|k_value_ranges|k_histogram|k_tables|k_i32}" "$out/v$v/s_kernel_stats.csv" | cut -d, -f1-4 | sed 's/dmi::(anonymous namespace):://'
done
