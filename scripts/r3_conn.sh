#!/bin/bash
# round 3: device connectivity tables — tests, then the end-to-end trace of the 10M-triangle workload
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_device_conn.py -x -q > gpurun_out/r3/conn_tests.log 2>&1
tail -5 gpurun_out/r3/conn_tests.log
python scripts/e2e_trace.py 2>&1 | grep -v "attribute . small" > gpurun_out/r3/e2e_trace.log
tail -30 gpurun_out/r3/e2e_trace.log
