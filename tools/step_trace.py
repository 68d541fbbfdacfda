"""Ordered list of the libofx entry-point calls of ONE eager denoising step with their HIP-event durations and shape
tags (the per-class view of the same data is bench.py's `roofline_tail`):

    python tools/step_trace.py [--workload hr] [--batch 8] --out gpurun_out/step_trace_hr.json
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from octfusion_amd import _lib, ops

ap = argparse.ArgumentParser()
ap.add_argument('--workload', default='hr')
ap.add_argument('--batch', type=int, default=None)
ap.add_argument('--out', default=None)
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--tile', type=int, default=0, help='force the planes-kernel geometry (2 / 4; 0 = automatic)')
a = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
_lib.require_device()
ops.set_precision('fp16x3')
_lib.call('ofx_set_gconv2_tile', a.tile)
w = bench.WORKLOADS[a.workload]
wl = bench.Workload(a.workload, a.batch or w['batch'], dev, 0)
wl.run(0, 4)
torch.cuda.synchronize()
rec = []
_lib.PROFILE = rec
t = bench.timed(lambda: wl.run(4, a.steps))
_lib.PROFILE = None
rows = [{'call': n.replace('ofx_', ''), 'ms': e0.elapsed_time(e1), 'meta': list(m) if m else None} for n, e0, e1, m in rec]
per = len(rows) // a.steps
out = {'workload': a.workload, 'batch': wl.batch, 'steps': a.steps, 'eager_ms_per_step': 1e3 * t / a.steps,
       'calls_per_step': per, 'last_step': rows[-per:]}
agg = {}
for r in rows:
    k = r['meta'][0] if r['meta'] else r['call']
    c = agg.setdefault(k, [0, 0.0])
    c[0] += 1
    c[1] += r['ms']
out['per_class_ms_per_step'] = {k: {'calls': v[0] / a.steps, 'ms': v[1] / a.steps} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
if a.out:
    json.dump(out, open(a.out, 'w'), indent=1)
print(json.dumps({k: out[k] for k in ('workload', 'batch', 'eager_ms_per_step', 'calls_per_step', 'per_class_ms_per_step')}))
