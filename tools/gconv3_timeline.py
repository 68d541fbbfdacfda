"""Per-block, per-piece timeline of the persistent planes GraphConv (csrc/ofx_gemm3.hip): where a block's time goes.
Needs the profiling build:  python -m octfusion_amd.build --ablation;  OFX_LIB=octfusion_amd/libofx_ablation.so python tools/gconv3_timeline.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from octfusion_amd import _lib, ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_small

dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
B = int(os.environ.get('G3_BATCH', '8'))
doc = DualOctree(split2octree_small(synthetic.shell6_split(B, jitter=True).to(dev), 6, 4))
ops.PLANES_MIN_TILES = 1
TILES = [int(v) for v in os.environ.get('G3_TILES', '4,2').split(',')]
SHAPES = [(6, 128, 128, True), (6, 128, 128, False), (6, 384, 128, True), (5, 256, 256, True), (5, 512, 512, True)]
_lib.call('ofx_set_gconv_persistent', 1)
for (d, cin, cout, epi) in SHAPES:
    for tile in TILES:
        _lib.call('ofx_set_gconv2_tile', tile)
        N = doc.csr(d)[2]
        conv = M.GraphConv(cin, cout, 7, 7, d - 1).to(dev)
        conv.emit_stats = epi
        gn = M.DualOctreeGroupNorm(cin).to(dev)
        xp = gn(torch.randn(N, cin, device=dev), doc, d, act='silu', planes=2)       # planes + aux rows: no pre-pass launch
        emb = torch.randn(B, cout, device=dev) if epi else None
        res = torch.randn(N, cout, device=dev) if epi else None

        def run():
            with ops.stats_scope(dev):
                return conv(xp, doc, d, emb=emb, res=res)
        for _ in range(3):
            run()
        nblk = 1024
        buf = torch.zeros(nblk * 16, dtype=torch.int64, device=dev)
        _lib.call('ofx_set_gconv2_debug', buf.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        _lib.call('ofx_set_gconv2_debug', None)
        if os.environ.get('G3_RAW'):
            os.makedirs(os.environ['G3_RAW'], exist_ok=True)
            torch.save(dict(stamps=buf.view(nblk, 16).cpu(), d=d, cin=cin, cout=cout, epi=epi, tile=tile),
                       os.path.join(os.environ['G3_RAW'], 'stamps_d%d_%d_%d_%d_t%d.pt' % (d, cin, cout, int(epi), tile)))
        t = buf.view(nblk, 16).cpu().double()
        t = t[t[:, 0] > 0]
        G = t.shape[0]
        t0 = t[:, 0].min()
        nkt = 7 * (cin // 32) + (7 * (d - 1) + 31) // 32
        print('d%d %d->%d epi=%s tile %d rows: %d blocks, nkt %d, launch %.1f us (events, incl. statistics reduce)'
              % (d, cin, cout, epi, tile * 64, G, nkt, e0.elapsed_time(e1) * 1e3))
        print('   start spread %.0f ticks; prologue (table + first two k tiles) mean %.0f max %.0f; block total mean %.0f max %.0f; chip span %.0f'
              % ((t[:, 0] - t0).max(), (t[:, 1] - t[:, 0]).mean(), (t[:, 1] - t[:, 0]).max(), (t[:, 3] - t[:, 0]).mean(),
                 (t[:, 3] - t[:, 0]).max(), t[:, 3].max() - t0))
        npc = t[:, 2].long().clamp(max=6)
        kl, ep = [], []
        for b in range(G):
            prev = t[b, 1]
            for p in range(int(npc[b])):
                kl.append(float(t[b, 4 + 2 * p] - prev))
                ep.append(float(t[b, 5 + 2 * p] - t[b, 4 + 2 * p]))
                prev = t[b, 5 + 2 * p]
        kl, ep = torch.tensor(kl), torch.tensor(ep)
        print('   pieces per block mean %.2f; k-loop per piece mean %.0f (sum per block %.0f); result (epilogue / publish) per piece mean %.0f max %.0f (sum per block %.0f)'
              % (npc.double().mean(), kl.mean(), kl.sum() / G, ep.mean(), ep.max(), ep.sum() / G))
        # full tiles only: k-loop ticks per k tile
        full = kl[(kl > 0.8 * kl.max())]
        if len(full):
            print('   longest pieces (whole tiles): %.0f ticks = %.0f per k tile' % (full.mean(), full.mean() / nkt))
_lib.call('ofx_set_gconv2_tile', 0)
