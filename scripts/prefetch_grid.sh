#!/bin/bash
# The sweep with its next chunk's fan rows prefetched (-DDMI_SWEEP_PREFETCH build as libdraco_mi_ab1.so) against the grid cap: more chunks
# per block = more iterations that have a predecessor to prefetch from.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "" _ab1; do for cap in ${CAPS:-8192 4096 2048 1792}; do
  out=$root/gpurun_out/pfgrid/v${lib}_$cap; mkdir -p "$out"
  DMI_FUSED_GRID=$cap DMI_LIBRARY=$root/draco-oxide_amd/libdraco_mi$lib.so rocprofv3 --kernel-trace --stats -d "$out" -o s --output-format csv -- python3 "$root/scripts/sweep_ablation.py" 2236 ${SUBSETS:-3} > "$out.log" 2>&1
  echo "lib '$lib' cap $cap: $(grep -E 'k_predict' "$out/s_kernel_stats.csv" | cut -d, -f1,4 | sed 's/"dmi::(anonymous namespace):://; s/(dmi::FusedArgs)"//' | tr '\n' ' ')"
done; done
