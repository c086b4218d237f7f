#!/usr/bin/env python3
"""bench.py's transcode_regime / batch_regime on their own: python scripts/transcode_time.py [n_files]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
print(json.dumps(bench.transcode_regime(n), indent=1))
if len(sys.argv) > 2:
    print(json.dumps(bench.batch_regime(), indent=1))
