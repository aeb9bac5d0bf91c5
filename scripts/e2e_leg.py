#!/usr/bin/env python
"""bench.py's secondary.e2e leg on its own: e2e_leg.py -> one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

torch.cuda.set_device(0)
print(json.dumps(bench.e2e_bench("cuda:0")))
