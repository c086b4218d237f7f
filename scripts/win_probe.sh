for w in 0 1; do
  if [ $w = 1 ]; then export DMI_FUSED_WINDOWS=1; else unset DMI_FUSED_WINDOWS; fi
  echo -n "windows=$w  "
  python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batch --no-scopes 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.readline()); print('predict_ms', b['stages_ms']['predict_ms'], 'value', b['value'])"
done
DMI_FUSED_WINDOWS=1 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
