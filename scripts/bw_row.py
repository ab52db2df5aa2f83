"""One row of bench.py's bandwidth_rooflines in a process of its own (scripts/profile_bw_rows.sh runs it under rocprofv3 --pmc:
the HBM bytes of exactly that row's kernels)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

rows = bench.bandwidth_rooflines(torch.device("cuda:0"), only=sys.argv[1])
print(json.dumps({k: v for k, v in rows.items() if isinstance(v, dict)}))
