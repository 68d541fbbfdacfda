import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from octfusion_amd import _lib, ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_small
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
B = 8
doc = DualOctree(split2octree_small(synthetic.shell6_split(B, jitter=False).to(dev), 6, 4))
ops.PLANES_MIN_TILES = 1
for d, cin, cout in [(5, 512, 512), (5, 256, 256), (4, 512, 512)]:
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, d - 1).to(dev); conv.emit_stats = False
    xp = ops.planes_split(torch.randn(N, cin, device=dev), 2)
    emb = torch.randn(B, cout, device=dev); res = torch.randn(N, cout, device=dev)
    nkt = 7 * cin // 32 + ((7 * (d - 1) + 31) // 32)
    for tile in (4, 8):
        _lib.call('ofx_set_gconv2_tile', tile)
        for _ in range(3): conv(xp, doc, d, emb=emb, res=res)
        rows = 256
        nblk = ((N + rows - 1) // rows) * (cout // (128 if tile == 4 else 256))
        buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
        _lib.call('ofx_set_gconv2_debug', buf.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); conv(xp, doc, d, emb=emb, res=res); e1.record(); torch.cuda.synchronize()
        _lib.call('ofx_set_gconv2_debug', None)
        t = buf.view(nblk, 8).cpu().double()
        seg = [(t[:, i + 1] - t[:, i]).mean() for i in range(4)]
        print('d%d %d->%d N=%d tile%d: %d blocks, %.1f us; table %.0f first-dma %.0f k-loop %.0f (%.0f per k-step, nkt %d) epilogue %.0f'
              % (d, cin, cout, N, tile, nblk, e0.elapsed_time(e1) * 1e3, seg[0], seg[1], seg[2], seg[2] / nkt, nkt, seg[3]))
    _lib.call('ofx_set_gconv2_tile', 0)
