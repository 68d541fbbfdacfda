"""A/B of the two launch shapes of the planes GraphConv on the layers of a bench workload (GPU box):
one tile per block (csrc/ofx_gemm2.hip, automatic geometry) vs persistent stream-K blocks (csrc/ofx_gemm3.hip) with
128- and 256-row tiles.  Per layer: microseconds per launch (HIP events, 20 launches), algorithmic TFLOP/s, fraction of
the bf16x3 roof, and max |difference| between the launch shapes.

    python tools/gconv3_ab.py [--batch 8] [--tree shell6|shell8] [--json out.json]
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


if args.tree == 'shell6':
    # the planes launches of one snet_uncond hr step (depth, cin, cout, launches per step)
    shapes = [(6, 128, 128, 4), (6, 256, 128, 2), (6, 384, 128, 1), (6, 256, 256, 1), (5, 128, 128, 1), (5, 128, 256, 1),
              (5, 256, 256, 4), (5, 512, 256, 1), (5, 768, 256, 1), (5, 384, 256, 1), (5, 512, 512, 1), (4, 256, 256, 1),
              (4, 256, 64, 1), (4, 128, 256, 1), (4, 512, 512, 2), (4, 768, 512, 1)]
else:
    shapes = [(8, 64, 64, 4), (8, 128, 64, 2), (8, 192, 64, 1), (7, 64, 64, 1), (7, 64, 128, 1), (7, 128, 128, 4),
              (7, 256, 128, 2), (7, 384, 128, 1), (6, 128, 128, 1), (6, 128, 256, 1), (6, 256, 256, 4)]
rows = []
tot = {}
for d, cin, cout, cnt in shapes:
    if d not in doc._csr:
        continue
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, d - 1).to(dev)
    conv.emit_stats = True
    xp = ops.planes_split(torch.randn(N, cin, device=dev), 2)
    emb = torch.randn(B, cout, device=dev)
    res = torch.randn(N, cout, device=dev)
    flops = 2.0 * N * 7 * (cin + d - 1) * cout
    out = dict(d=d, N=N, cin=cin, cout=cout, launches_per_step=cnt)
    ys = {}
    for name, pers, tile in (('tile', 0, 0), ('pk2', 2, 2), ('pk4', 2, 4), ('hy4f', 3, 4), ('hy4', 1, 4)):       # pk*: pure stream-K; hy4: rounds + region (hy4f: round-1 share snapping)
        _lib.call('ofx_set_gconv_persistent', pers)
        _lib.call('ofx_set_gconv2_tile', tile)

        def run():
            with ops.stats_scope(dev):
                return conv(xp, doc, d, emb=emb, res=res)
        ys[name] = run().clone()
        us = timeit(run, args.iters)
        out[name + '_us'] = us
        out[name + '_frac'] = flops / us / 1e6 / PEAK
        tot[name] = tot.get(name, 0.0) + us * cnt
    out['pk2_vs_tile'] = float((ys['pk2'] - ys['tile']).abs().max() / ys['tile'].abs().max())
    out['pk4_vs_tile'] = float((ys['pk4'] - ys['tile']).abs().max() / ys['tile'].abs().max())
    out['hy4_vs_tile'] = float((ys['hy4'] - ys['tile']).abs().max() / ys['tile'].abs().max())
    out['hy4f_vs_tile'] = float((ys['hy4f'] - ys['tile']).abs().max() / ys['tile'].abs().max())
    torch.cuda.synchronize()
    out['sync_error'] = ops.sync_error(dev)
    rows.append(out)
    print(json.dumps(out))
    sys.stdout.flush()
_lib.call('ofx_set_gconv_persistent', 1)
_lib.call('ofx_set_gconv2_tile', 0)
best = sum(min(r['tile_us'], r['pk2_us'], r['pk4_us'], r['hy4_us'], r['hy4f_us']) * r['launches_per_step'] for r in rows)
print(json.dumps(dict(summary='us per step over the listed launches', **tot, best_per_layer=best)))
if args.json:
    json.dump(dict(rows=rows, totals=tot, best_per_layer=best), open(args.json, 'w'), indent=1)
