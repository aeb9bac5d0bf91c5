#!/usr/bin/env python
"""bench.py's secondary.inference leg on its own (2 warm-up + 5 timed volumes = 7 create_super_volume calls): one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

torch.cuda.set_device(0)
print(json.dumps(bench.inference_bench("cuda:0")))
