#!/usr/bin/env python
"""The small-shard table of bench.py (secondary.small_shards) on its own: small_shards_leg.py [tag] -> one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

torch.cuda.set_device(0)
out = bench.small_shards("cuda:0")
out["tag"] = sys.argv[1] if len(sys.argv) > 1 else ""
print(json.dumps(out))
