# the 10M-triangle call with the walks over 4·face + k ids and without: bash scripts/experiments/quad_ab.sh   (edit the variable name to DMI_NO_CLOSED for the closed-mesh loops)
for i in 1 2 3 4 5; do
for v in "" "1"; do
  if [ -n "$v" ]; then export DMI_NO_QUAD=1; else unset DMI_NO_QUAD; fi
  python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-batch --no-scopes --transcode-files 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('NO_QUAD=$v', d['value'], d['ms_per_step'], 'conn', round(s['connectivity_ms'],2), 'tables', round(s['tables_ms'],2), 'enc', round(s['total_ms'],2))"
done; done
