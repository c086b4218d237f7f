#!/bin/bash
# Is the fused predictor sweep bound by vector issue or by memory latency?  Unused dynamic LDS per block (DMI_FUSED_LDS) lowers the
# number of resident waves per SIMD without touching the code: a latency-bound kernel slows down in proportion, an issue-bound one
# does not until too few waves are left to keep the vector unit fed.
for lds in 0 16384 32768 40960 65536; do
  echo -n "DMI_FUSED_LDS=$lds  "
  DMI_FUSED_LDS=$lds python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batch --no-scopes 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.readline()); print('predict_ms', b['stages_ms']['predict_ms'], 'quantize_ms', b['stages_ms']['quantize_ms'])"
done
