#!/bin/bash
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_batch_prepare.py tests/test_gpu_device_conn.py -x -q > gpurun_out/r3/batch_tests.log 2>&1
tail -25 gpurun_out/r3/batch_tests.log
DMI_TRACE=1 python scripts/bench_batch.py 2>&1 | grep -v "small:" | tail -12 > gpurun_out/r3/bench_batch.log
cat gpurun_out/r3/bench_batch.log
