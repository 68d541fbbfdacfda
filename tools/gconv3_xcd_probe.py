"""A/B of the persistent planes GraphConv's tile order (ofx_set_gconv_xcd_contig): XCDs interleaved inside every round
(round 3) vs one contiguous tile range per XCD (round 6), per layer shape of the hr / feature workloads (GPU box).
    python tools/gconv3_xcd_probe.py [shell6|shell8] [--json out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octfusion_amd import _lib, ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_large, split2octree_small
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
tree = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 'shell6'
out = sys.argv[sys.argv.index('--json') + 1] if '--json' in sys.argv else None
B = 8
oc = split2octree_small(synthetic.shell6_split(B, jitter=True).to(dev), 6, 4)
if tree == 'shell8':
    x6, y6, z6, _ = oc.xyzb(6)
    oc = split2octree_large(oc, synthetic.shell8_split_large(x6, y6, z6), 6)
doc = DualOctree(oc)
ops.PLANES_MIN_TILES = 1
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
if tree == 'shell6':
    shapes = [(6, 128, 128, 4), (6, 256, 128, 2), (6, 384, 128, 1), (6, 256, 256, 1), (5, 256, 256, 4), (5, 512, 256, 1),
              (5, 768, 256, 1), (5, 384, 256, 1), (5, 512, 512, 1), (4, 512, 512, 2), (4, 256, 256, 1)]
else:
    shapes = [(8, 64, 64, 4), (8, 128, 64, 1), (8, 192, 64, 1), (8, 128, 128, 1), (7, 128, 128, 4), (7, 256, 128, 1),
              (7, 384, 128, 1), (7, 256, 256, 1), (6, 256, 256, 6), (6, 128, 128, 6), (6, 512, 256, 1)]
rows, tot = [], [0.0, 0.0]
for d, cin, cout, cnt in shapes:
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, d - 1).to(dev)
    gn = M.DualOctreeGroupNorm(cin).to(dev)
    x = torch.randn(N, cin, device=dev)
    hp = gn(x, doc, d, act='silu', planes=ops.planes_mode())
    res = torch.randn(N, cout, device=dev)
    r = dict(d=d, N=N, cin=cin, cout=cout, launches=cnt)
    ys = []
    for k in (0, 1, 0, 1):
        _lib.call('ofx_set_gconv_xcd_contig', k)
        def run():
            with ops.stats_scope(dev):
                return conv(hp, doc, d, res=res)
        y = run().clone()
        us = timeit(run)
        key = 'contig_us' if k else 'interleaved_us'
        r[key] = min(r.get(key, 1e9), us)
        ys.append(y)
    r['max_diff'] = float((ys[0] - ys[1]).abs().max() / ys[0].abs().max())
    r['gain'] = 1.0 - r['contig_us'] / r['interleaved_us']
    tot[0] += r['interleaved_us'] * cnt; tot[1] += r['contig_us'] * cnt
    rows.append(r); print(json.dumps(r), flush=True)
_lib.call('ofx_set_gconv_xcd_contig', 0)
print(json.dumps(dict(total_interleaved_us=tot[0], total_contig_us=tot[1], sync_error=ops.sync_error(dev))))
if out: json.dump(dict(rows=rows, totals=tot), open(out, 'w'), indent=1)
