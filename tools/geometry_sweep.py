"""Per-layer check of the planes GraphConv's automatic block geometry (persistent launch): every planes layer of a bench
workload with the automatic choice, forced 128-row tiles (tile 2) and forced 256-row tiles (tile 4), default precision,
emb + residual + fused statistics.  us per launch (HIP events), fraction of the three-term roof.

    python tools/geometry_sweep.py [--tree shell6|shell8] [--batch 8] [--json out.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from octfusion_amd import _lib, ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_large, split2octree_small

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--tree', default='shell6')
ap.add_argument('--json', default=None)
ap.add_argument('--iters', type=int, default=20)
args = ap.parse_args()
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
B = args.batch
oc = split2octree_small(synthetic.shell6_split(B, jitter=True).to(dev), 6, 4)
if args.tree == 'shell8':
    x6, y6, z6, _ = oc.xyzb(6)
    oc = split2octree_large(oc, synthetic.shell8_split_large(x6, y6, z6), 6)
doc = DualOctree(oc)
ops.PLANES_MIN_TILES = 1
PEAK = 2500.0 / 3.0
if args.tree == 'shell6':
    shapes = [(6, 128, 128, 4), (6, 256, 128, 2), (6, 384, 128, 1), (6, 256, 256, 1), (5, 128, 128, 1), (5, 128, 256, 1),
              (5, 256, 256, 4), (5, 512, 256, 1), (5, 768, 256, 1), (5, 384, 256, 1), (5, 512, 512, 1), (4, 256, 256, 1),
              (4, 256, 64, 1), (4, 128, 256, 1), (4, 512, 512, 2), (4, 768, 512, 1), (4, 64, 64, 1)]
else:
    shapes = [(8, 64, 64, 4), (8, 128, 64, 2), (8, 192, 64, 1), (7, 64, 64, 1), (7, 64, 128, 1), (7, 128, 128, 4),
              (7, 256, 128, 2), (7, 384, 128, 1), (6, 128, 128, 1), (6, 128, 256, 1), (6, 256, 256, 4)]


def timeit(fn, n):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


rows, tot = [], {}
mode = ops.planes_mode()
for d, cin, cout, cnt in shapes:
    if d not in doc._csr:
        continue
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, d - 1).to(dev)
    gn = M.DualOctreeGroupNorm(cin).to(dev)
    xp = gn(torch.randn(N, cin, device=dev), doc, d, act='silu', planes=mode)
    emb = torch.randn(B, cout, device=dev)
    res = torch.randn(N, cout, device=dev)
    flops = 2.0 * N * 7 * (cin + d - 1) * cout
    out = dict(d=d, N=N, cin=cin, cout=cout, launches_per_step=cnt)
    for name, tile in (('auto', 0), ('t2', 2), ('t4', 4)):
        _lib.call('ofx_set_gconv2_tile', tile)

        def run():
            with ops.stats_scope(dev):
                return conv(xp, doc, d, emb=emb, res=res)
        us = timeit(run, args.iters)
        out[name + '_us'] = round(us, 1)
        out[name + '_frac'] = round(flops / us / 1e6 / PEAK, 3)
        tot[name] = tot.get(name, 0.0) + us * cnt
    rows.append(out)
    print(json.dumps(out), flush=True)
_lib.call('ofx_set_gconv2_tile', 0)
best = sum(min(r['auto_us'], r['t2_us'], r['t4_us']) * r['launches_per_step'] for r in rows)
print(json.dumps(dict(summary='us per step over the listed launches', **{k: round(v, 1) for k, v in tot.items()}, best_per_layer=round(best, 1))))
if args.json:
    json.dump(dict(rows=rows, totals=tot, best_per_layer=best), open(args.json, 'w'), indent=1)
