"""256 x 256 / 4-wave kernel (ofx_set_gconv2_tile 8) against the 256 x 128 / 8-wave one (tile 4) on the cout >= 256
bench layers: bit-equality of the outputs (same accumulation order), fused statistics, time per launch, timeline."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from octfusion_amd import _lib, ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_small

dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
B = 8
doc = DualOctree(split2octree_small(synthetic.shell6_split(B, jitter=True).to(dev), 6, 4))
ops.PLANES_MIN_TILES = 1
QUICK = len(sys.argv) > 1 and sys.argv[1] == 'quick'


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


shapes = [(5, 256, 256), (5, 512, 512), (5, 512, 256), (5, 768, 256), (6, 128, 256), (4, 512, 512), (4, 256, 256), (5, 128, 256)]
if QUICK:
    shapes = shapes[:2]
for d, cin, cout in shapes:
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, d - 1).to(dev)
    xp = ops.planes_split(torch.randn(N, cin, device=dev), 2)
    emb = torch.randn(B, cout, device=dev)
    res = torch.randn(N, cout, device=dev)
    flops = 2.0 * N * 7 * (cin + d - 1) * cout
    out = {}
    for tile in (4, 8):
        _lib.call('ofx_set_gconv2_tile', tile)
        conv.emit_stats = True
        y = conv(xp, doc, d, emb=emb, res=res)
        st = ops.get_stats(y).clone()
        torch.cuda.synchronize()
        conv.emit_stats = False
        t = timeit(lambda: conv(xp, doc, d, emb=emb, res=res))
        out[tile] = (y.clone(), st, t)
    _lib.call('ofx_set_gconv2_tile', 0)
    same = torch.equal(out[4][0], out[8][0])
    err = float((out[4][0] - out[8][0]).abs().max() / out[4][0].abs().max())
    serr = float((out[4][1] - out[8][1]).abs().max() / out[4][1].abs().max())
    print('d%d %d->%d N=%d: bit-equal %s (max rel diff %.2e), stats rel diff %.2e; tile4 %.1f us (%.0f TF/s)  tile8 %.1f us (%.0f TF/s)  ratio %.3f'
          % (d, cin, cout, N, same, err, serr, out[4][2], flops / out[4][2] / 1e6, out[8][2], flops / out[8][2] / 1e6, out[8][2] / out[4][2]))
    sys.stdout.flush()
