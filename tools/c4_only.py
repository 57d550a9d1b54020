#!/usr/bin/env python3
"""Development aid: the decode leg of bench.py alone (an n-hour -8 stream decoded at 1, 2, 4, ... hour batches)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

hours = int(sys.argv[1]) if len(sys.argv) > 1 else 8
base, _ = bench.build_workload(60, 1)
res = bench.config_c4(torch.device("cuda", 0), 0, base, hours)
print(json.dumps(res["by_batch"]))
