#!/bin/bash
# quantize+predict pass of the resident workload under the tile-sorted quantize gather (env switch read at job creation)
n=${1:-2236}
shift
for v in "" "$@"; do
  echo "== ${v:-default}"
  env $v python scripts/resident_steps.py $n 5 2>&1 | tail -1 | cut -c1-120
done
