"""Scheduling variants of the planes kernel (1 = DMA interleaved by sched_group_barrier, 5 = LDS reads and DMA requests
spliced between the MFMAs) x tile geometry, per bench layer: bit-equality of the outputs and us per launch (GPU box)."""
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
    return 1e3 * e0.elapsed_time(e1) / n
for d, cin, cout in [(6, 128, 128), (6, 384, 128), (6, 256, 256), (5, 256, 256), (5, 512, 512), (4, 512, 512), (6, 128, 64), (6, 64, 64)]:
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, d - 1).to(dev)
    conv.emit_stats = False
    xp = ops.planes_split(torch.randn(N, cin, device=dev), 2)
    res = torch.randn(N, cout, device=dev)
    emb = torch.randn(8, cout, device=dev)
    ref, row = None, []
    for tile in (4, 2):
        for variant in [int(v) for v in os.environ.get('G2_VARIANTS', '1,5').split(',')]:
            _lib.call('ofx_set_gconv2_tile', tile)
            _lib.call('ofx_set_gconv2_variant', variant)
            y = conv(xp, doc, d, emb=emb, res=res)
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            same = bool(torch.equal(y, ref))
            t = timeit(lambda: conv(xp, doc, d, emb=emb, res=res))
            row.append('WM%d v%d %.1f us%s' % (tile, variant, t, '' if same else ' MISMATCH'))
    print('d%d %4d->%3d: ' % (d, cin, cout) + '   '.join(row))
_lib.call('ofx_set_gconv2_tile', 0)
_lib.call('ofx_set_gconv2_variant', 5)
