#!/usr/bin/env python3
"""The reference's Criterion workloads alone (bench.py's criterion leg: 2000x1800 pattern, six Encoder configurations, pageable /
registered / register-cache buffers), one JSON line per configuration."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")
res, _ = bench.criterion_workloads(b, synth, 0)
for k, v in res.items():
    if isinstance(v, dict):
        print(json.dumps({"workload": k, **v}))
