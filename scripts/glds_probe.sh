#!/bin/bash
# DMI_SWEEP_GLDS experiment: the sweep with its level-1 data staged global → LDS by DMA one chunk ahead, against the default build.
# Both libraries: byte parity on the 1M-triangle test, then kernel durations of the 10M-triangle workload under rocprofv3.
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/glds
mkdir -p "$out"
cd "$root"
for v in "" ${VARIANTS:-_glds}; do
  lib=$root/draco-oxide_amd/libdraco_mi$v.so
  [ -f "$lib" ] || continue
  echo "== variant '$v': parity"
  DMI_LIBRARY=$lib timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "synthetic_drc_bit_exact or one_million or fan_rows or seams" 2>&1 | tail -2
done
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for v in "" ${VARIANTS:-_glds}; do
  lib=$root/draco-oxide_amd/libdraco_mi$v.so
  [ -f "$lib" ] || continue
  DMI_LIBRARY=$lib rocprofv3 --kernel-trace --stats -d "$out/v$v$rep" -o s --output-format csv -- python3 "$root/scripts/sweep_ablation.py" 2236 ${SUBSETS:-3} > "$out/v$v$rep.log" 2>&1
  echo "variant '$v' rep $rep"; grep -E "k_predict_packed|k_seq_quantize" "$out/v$v$rep/s_kernel_stats.csv" | cut -d, -f1-5
done
done
