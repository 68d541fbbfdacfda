"""Sweep of the start offset between the two co-resident blocks of the 128-row planes kernel (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octfusion_amd import _lib, ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_small
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
doc = DualOctree(split2octree_small(synthetic.shell6_split(8).to(dev), 6, 4))
ops.PLANES_MIN_TILES = 1
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for d, cin, cout in [(6, 128, 128), (6, 384, 128), (5, 256, 256), (5, 512, 512), (4, 512, 512)]:
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, d - 1).to(dev)
    conv.emit_stats = False
    xp = ops.planes_split(torch.randn(N, cin, device=dev), 2)
    res = torch.randn(N, cout, device=dev)
    emb = torch.randn(8, cout, device=dev)
    row = []
    _lib.call('ofx_set_gconv2_tile', 4)
    row.append(('wm4', timeit(lambda: conv(xp, doc, d, emb=emb, res=res))))
    _lib.call('ofx_set_gconv2_tile', 2)
    for st in (0, 400, 800, 1100, 1500, 2000):
        _lib.call('ofx_set_gconv2_stagger', st)
        row.append(('st%d' % st, timeit(lambda: conv(xp, doc, d, emb=emb, res=res))))
    print('d%d %d->%d N=%d: ' % (d, cin, cout, N) + '  '.join('%s %.3f' % r for r in row))
