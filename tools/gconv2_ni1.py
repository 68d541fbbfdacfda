"""64-column-tile layers of the feature net (depth 8 / 7, cout 64) on the shell-8 tree: two-slot vs three-slot
assumption for the start offsets and the table-prefetch distance (ofx_set_gconv2_prefetch 3 vs 1), and the block
start pattern of the first 768 blocks (do three blocks per CU really start together?)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from octfusion_amd import _lib, ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_large, split2octree_small

dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
B = int(os.environ.get('B', '8'))
oc = split2octree_small(synthetic.shell6_split(B, jitter=True).to(dev), 6, 4)
x6, y6, z6, _ = oc.xyzb(6)
oc = split2octree_large(oc, synthetic.shell8_split_large(x6, y6, z6), 6)
doc = DualOctree(oc)
ops.PLANES_MIN_TILES = 1


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for d, cin, cout in [(8, 64, 64), (8, 128, 64), (8, 192, 64), (7, 64, 64)]:
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, d - 1).to(dev)
    conv.emit_stats = False
    xp = ops.planes_split(torch.randn(N, cin, device=dev), 2)
    emb = torch.randn(B, cout, device=dev)
    res = torch.randn(N, cout, device=dev)
    nblk = (N + 127) // 128
    line = 'd%d %d->%d N=%d (%d blocks):' % (d, cin, cout, N, nblk)
    ys = []
    for mode in (3, 1, 3, 1):
        _lib.call('ofx_set_gconv2_prefetch', mode)
        ys.append(conv(xp, doc, d, emb=emb, res=res).clone())
        line += '  %s %.1f us' % ('2-slot' if mode == 3 else '3-slot', timeit(lambda: conv(xp, doc, d, emb=emb, res=res)))
    print(line, ' bit-equal', torch.equal(ys[0], ys[1]))
    _lib.call('ofx_set_gconv2_prefetch', 1)
    _lib.call('ofx_set_gconv2_stagger', 0)
    buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
    _lib.call('ofx_set_gconv2_debug', buf.data_ptr())
    conv(xp, doc, d, emb=emb, res=res)
    torch.cuda.synchronize()
    _lib.call('ofx_set_gconv2_debug', None)
    _lib.call('ofx_set_gconv2_stagger', 1100)
    t = buf.view(nblk, 8).cpu().double()
    st = t[:, 0] - t[:, 0].min()
    dur = (t[:, 4] - t[:, 0])[:768].mean()
    print('   without offsets: start ticks of blocks 0-255 %.0f, 256-511 %.0f, 512-767 %.0f, 768-1023 %.0f (mean); block duration %.0f'
          % (st[:256].mean(), st[256:512].mean(), st[512:768].mean(), st[768:1024].mean(), dur))
