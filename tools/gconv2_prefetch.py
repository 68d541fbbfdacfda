"""A/B of the neighbour-table prefetch (ofx_set_gconv2_prefetch) on the bench layers: time per launch, the table
segment of the per-block timeline for first-round and later-round blocks, and bit-equality of the outputs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from octfusion_amd import _lib, ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_small

dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
doc = DualOctree(split2octree_small(synthetic.shell6_split(8, jitter=True).to(dev), 6, 4))
ops.PLANES_MIN_TILES = 1


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for d, cin, cout in [(6, 128, 128), (6, 384, 128), (6, 256, 256), (5, 256, 256), (5, 512, 256), (5, 768, 256), (4, 512, 512)]:
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, d - 1).to(dev)
    conv.emit_stats = False
    xp = ops.planes_split(torch.randn(N, cin, device=dev), 2)
    emb = torch.randn(8, cout, device=dev)
    res = torch.randn(N, cout, device=dev)
    wm = 2 if cout <= 128 else 4
    resident = 512 if wm == 2 else 256
    nblk = ((N + 64 * wm - 1) // (64 * wm)) * ((cout + 127) // 128)
    out = {}
    for pf in (0, 1, 0, 1):
        _lib.call('ofx_set_gconv2_prefetch', pf)
        y = conv(xp, doc, d, emb=emb, res=res).clone()
        t = timeit(lambda: conv(xp, doc, d, emb=emb, res=res))
        buf = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
        _lib.call('ofx_set_gconv2_debug', buf.data_ptr())
        conv(xp, doc, d, emb=emb, res=res)
        torch.cuda.synchronize()
        _lib.call('ofx_set_gconv2_debug', None)
        tt = buf.view(nblk, 8).cpu().double()
        tab = tt[:, 1] - tt[:, 0]
        out.setdefault(pf, []).append((t, float(tab[:resident].mean()), float(tab[resident:].mean()) if nblk > resident else float('nan'), y))
    same = torch.equal(out[0][0][3], out[1][0][3])
    f = lambda pf: ' / '.join('%.1f us (table %0.f first round, %.0f later)' % r[:3] for r in out[pf])     # noqa: E731
    print('d%d %d->%d (%d blocks, %d resident) bit-equal %s\n   off: %s\n   on : %s' % (d, cin, cout, nblk, resident, same, f(0), f(1)))
_lib.call('ofx_set_gconv2_prefetch', 1)
