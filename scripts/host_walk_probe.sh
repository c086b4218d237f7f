#!/bin/bash
# Host serial walks (Edgebreaker traversal, sequencer) of the 10M-triangle workload under the adjacent-line prefetch (DMI_PF) and
# transparent huge pages (DMI_NO_THP) switches; run on the GPU box:  gpurun -- bash scripts/host_walk_probe.sh
out=gpurun_out/r3/host_walk_probe2.log
mkdir -p gpurun_out/r3
{
  echo "thp: $(cat /sys/kernel/mm/transparent_hugepage/enabled) defrag: $(cat /sys/kernel/mm/transparent_hugepage/defrag)"
  for cfg in "16 0 0" "16 0 1" "32 0 0" "8 0 0" "0 0 0"; do
    set -- $cfg
    echo "== DMI_PF=$1 NO_THP=$2 NOFLAGS=$3"
    if [ "$2" = 1 ]; then export DMI_NO_THP=1; else unset DMI_NO_THP; fi
    if [ "$3" = 1 ]; then export DMI_PF_NOFLAGS=1; else unset DMI_PF_NOFLAGS; fi
    DMI_PF=$1 DMI_TRACE=1 DMI_TRACE_TABLES=1 python scripts/conn_time.py 2236 4 2>&1 | grep -E "universal table|Edgebreaker of|host connectivity|encode_connectivity" | tail -4
  done
  unset DMI_NO_THP DMI_PF_NOFLAGS DMI_PF
  echo "== e2e trace (defaults)"
  python scripts/e2e_trace.py 2>&1 | grep -v "attribute . small" | tail -40
} > $out 2>&1
