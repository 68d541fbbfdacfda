"""The stand-alone segment-mean gather (the reference's col_data = scatter_mean(x[col]), modules.py:208-210) on the
bench tree, for a rocprofv3 --kernel-trace --stats row of gather_mean_kernel next to bench.py's HIP-event figure."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
_, doc, _ = bench.build_tree('shell6', 8, dev)
print(json.dumps(bench.gather_microbench(doc, dev)))
