import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from octfusion_amd import _lib, ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_large, split2octree_small
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
B = 8
oc = split2octree_small(synthetic.shell6_split(B, jitter=True).to(dev), 6, 4)
x6, y6, z6, _ = oc.xyzb(6)
oc = split2octree_large(oc, synthetic.shell8_split_large(x6, y6, z6), 6)
doc = DualOctree(oc); ops.PLANES_MIN_TILES = 1
def timeit(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for d, cin, cout in [(8, 64, 64), (8, 128, 64), (8, 192, 64), (8, 128, 128), (7, 128, 128), (7, 256, 128)]:
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, d - 1).to(dev); conv.emit_stats = False
    xp = ops.planes_split(torch.randn(N, cin, device=dev), 2)
    emb = torch.randn(B, cout, device=dev); res = torch.randn(N, cout, device=dev)
    line = 'd%d %d->%d:' % (d, cin, cout)
    for tile in (2, 4, 2, 4):
        _lib.call('ofx_set_gconv2_tile', tile)
        line += '  tile%d %.1f us' % (tile, timeit(lambda: conv(xp, doc, d, emb=emb, res=res)))
    _lib.call('ofx_set_gconv2_tile', 0)
    print(line)
