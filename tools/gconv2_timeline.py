"""Per-block timeline of the planes GraphConv kernel (ofx_set_gconv2_debug): where a block's time goes
(table build / first DMA / k-loop / epilogue) and how the blocks are spread over the launch.  GPU box only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from octfusion_amd import _lib, ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_small

dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
doc = DualOctree(split2octree_small(synthetic.shell6_split(8).to(dev), 6, 4))
ops.PLANES_MIN_TILES = 1
import itertools
VARIANTS = [int(v) for v in os.environ.get("G2_VARIANTS", "5").split(",")]
TILES = [int(v) for v in os.environ.get("G2_TILES", "2,4").split(",")]
for (d, cin, cout, epi), variant, tile in itertools.product([(6, 128, 128, True), (6, 128, 128, False), (6, 384, 128, True), (5, 256, 256, True)], VARIANTS, TILES):
    _lib.call("ofx_set_gconv2_variant", variant)
    _lib.call("ofx_set_gconv2_tile", tile)
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, d - 1).to(dev)
    conv.emit_stats = False
    xp = ops.planes_split(torch.randn(N, cin, device=dev), 2)
    emb = torch.randn(8, cout, device=dev) if epi else None
    res = torch.randn(N, cout, device=dev) if epi else None
    for _ in range(3):
        conv(xp, doc, d, emb=emb, res=res)
    nblk = ((N + 64 * tile - 1) // (64 * tile)) * ((cout + 127) // 128)
    buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
    _lib.call('ofx_set_gconv2_debug', buf.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    conv(xp, doc, d, emb=emb, res=res)
    e1.record()
    torch.cuda.synchronize()
    _lib.call('ofx_set_gconv2_debug', None)
    t = buf.view(nblk, 8).cpu().double()
    t0 = t[:, 0].min()
    seg = [(t[:, i + 1] - t[:, i]) for i in range(4)]
    span = (t[:, 4].max() - t0)
    print('d%d %d->%d epi=%s variant %d tile %d: blocks %d, launch %.1f us (events)' % (d, cin, cout, epi, variant, tile * 64, nblk, e0.elapsed_time(e1) * 1e3))
    for name, s in zip(('table', 'first-dma', 'k-loop', 'epilogue'), seg):
        print('   %-10s mean %8.0f  min %8.0f  max %8.0f ticks' % (name, s.mean(), s.min(), s.max()))
    start = t[:, 0] - t0
    order = torch.argsort(start)
    print('   block start ticks: first %.0f, 256th %.0f, 512th %.0f, last %.0f' % (
        start[order[0]], start[order[min(255, nblk - 1)]], start[order[min(511, nblk - 1)]], start[order[-1]]))
    print('   block end ticks: min %.0f max %.0f' % ((t[:, 4] - t0).min(), (t[:, 4] - t0).max()))
