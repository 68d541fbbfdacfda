"""Single launch (forced 128- or 256-row geometry) vs the automatic bulk + remainder split, per bench layer (GPU box)."""
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
for d, cin, cout in [(6, 128, 128), (6, 256, 128), (6, 384, 128), (6, 256, 256), (5, 256, 256), (5, 384, 256), (5, 768, 256), (5, 512, 512), (4, 512, 512), (4, 256, 256)]:
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, d - 1).to(dev)
    conv.emit_stats = True
    xp = ops.planes_split(torch.randn(N, cin, device=dev), 2)
    res = torch.randn(N, cout, device=dev)
    emb = torch.randn(8, cout, device=dev)
    out = []
    ref = None
    for tile in (4, 2, 0):
        _lib.call('ofx_set_gconv2_tile', tile)
        with ops.stats_scope(dev):
            y = conv(xp, doc, d, emb=emb, res=res)
            st = ops.get_stats(y).clone()
        if ref is None:
            ref = (y.clone(), st)
        else:
            assert torch.equal(y, ref[0]), 'output differs between geometries'
            assert float((st - ref[1]).abs().max() / ref[1].abs().max()) < 1e-9
        def run():
            with ops.stats_scope(dev):
                conv(xp, doc, d, emb=emb, res=res)
        out.append(timeit(run))
    fl = 2.0 * N * 7 * (cin + d - 1) * cout
    print('d%d %4d->%3d N=%6d: WM4 %.1f us  WM2 %.1f us  auto(split) %.1f us  (%.0f TF/s)' % (d, cin, cout, N, out[0], out[1], out[2], fl / out[2] / 1e6))
