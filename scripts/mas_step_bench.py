"""The captured MAS training step alone (bench.py::mas_leg), for kernel statistics: python scripts/mas_step_bench.py [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "kan-tts_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import bench  # noqa: E402
import torch_oracle as O  # noqa: E402

import kantts._hip as hip  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
print(json.dumps(bench.mas_leg(hip, O.sambert_config(tiny=False), "bf16", "cuda", steps=steps, warmup=3)))
