"""A/B of a planes-kernel knob on a whole bench workload in ONE process (same box, same clocks): alternating eager
runs of K steps.   python tools/feature_ab.py [workload] [knob] [a] [b]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from octfusion_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else 'feature'
knob = sys.argv[2] if len(sys.argv) > 2 else 'ofx_set_gconv2_tile'
va, vb = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1, 0)
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
wl = bench.Workload(name, bench.WORKLOADS[name]['batch'] if 'batch' in bench.WORKLOADS[name] else 8, dev, 0)
wl.run(0, 3)
torch.cuda.synchronize()
K = 8
for rep in range(3):
    for v in (va, vb):
        _lib.call(knob, v)
        wl.run(0, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wl.run(0, K)
        torch.cuda.synchronize()
        print('%s %s(%d): %.3f ms/step' % (name, knob, v, 1e3 * (time.perf_counter() - t0) / K))
