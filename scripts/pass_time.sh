#!/bin/bash
# quantize+predict pass of the resident 10M workload, a few variants (env switches read at job creation)
for v in "" "DMI_NO_VALUE_OCT=1"; do
  echo "== ${v:-default}"
  env $v python scripts/resident_steps.py 2236 8 2>&1 | tail -1
done
