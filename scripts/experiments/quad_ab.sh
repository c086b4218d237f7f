# the 10M-triangle call with the walks over 4·face + k ids (default for meshes none of whose attributes needs a corner table of its own) and without
for i in 1 2 3; do
for v in "" "1"; do
  if [ -n "$v" ]; then export DMI_NO_QUAD=1; else unset DMI_NO_QUAD; fi
  python bench.py --steps 7 --warmup 2 --no-cpu-baseline --no-batch --no-scopes --transcode-files 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('NO_QUAD=$v', d['value'], d['ms_per_step'], 'conn', round(s['connectivity_ms'],2), 'tables', round(s['tables_ms'],2), 'enc', round(s['total_ms'],2))"
done; done
