# the 10M-triangle call under a switch of the serial walks and without, alternating: bash scripts/experiments/quad_ab.sh [DMI_NO_QUAD|DMI_NO_CLOSED] [pairs]
VAR=${1:-DMI_NO_QUAD}; N=${2:-3}
for i in $(seq $N); do
for v in "" "1"; do
  if [ -n "$v" ]; then export $VAR=1; else unset $VAR; fi
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-batch --no-scopes --transcode-files 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('$VAR=$v', d['value'], d['ms_per_step'], 'conn', round(s['connectivity_ms'],2), 'tables', round(s['tables_ms'],2), 'enc', round(s['total_ms'],2))"
done; done
