#!/bin/bash
# Kernel durations of the full sweep for throwaway builds of the library with parts of the sweep compiled out (-DDMI_ABLATE=n).
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/ablate_var
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for v in "" _ab1 _ab2 _ab3 _ab4 _ab5 _ab6 _ab7 _ab8 _ab9 _ab10 _ab11; do
  lib=$root/draco-oxide_amd/libdraco_mi$v.so
  [ -f "$lib" ] || continue
  DMI_LIBRARY=$lib rocprofv3 --kernel-trace --stats -d "$out/v$v" -o s --output-format csv -- python3 "$root/scripts/sweep_ablation.py" 2236 ${SUBSETS:-1} > "$out/v$v.log" 2>&1
  echo "variant '$v'"; grep -E "k_predict|k_seq_quantize|k_value_ranges\"|k_pred_" "$out/v$v/s_kernel_stats.csv" | cut -d, -f1-5
done
