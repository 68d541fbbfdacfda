"""Input GraphConv (3 / 8 -> 64 / 128 channels): CSR-walking launch (round 5) vs the table-driven persistent pipelined
launch (round 6), on the hr (shell-6 x 8, depth 6) and feature (shell-8 x 8, depth 8) trees (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octfusion_amd import ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_large, split2octree_small
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
ops.NARROW_IN_TAB_MIN_ROWS = 0
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
oc = split2octree_small(synthetic.shell6_split(8).to(dev), 6, 4)
docs = {'shell6x8': DualOctree(oc)}
x6, y6, z6, _ = oc.xyzb(6)
docs['shell8x8'] = DualOctree(split2octree_large(oc, synthetic.shell8_split_large(x6, y6, z6), 6))
for tag, d, cin, cout in (('shell6x8', 6, 3, 128), ('shell6x8', 6, 8, 128), ('shell8x8', 8, 3, 64), ('shell8x8', 7, 3, 64)):
    doc = docs[tag]
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, d - 1).to(dev)
    x = torch.randn(N, cin, device=dev)
    r = {}
    for tab in (False, True):
        ops.NARROW_IN_TAB = tab
        def run():
            with ops.stats_scope(dev):
                return conv(x, doc, d)
        r[tab] = (timeit(run), run().clone())
    ops.NARROW_IN_TAB = True
    diff = float((r[True][1] - r[False][1]).abs().max() / r[False][1].abs().max())
    print('%s depth %d, %d -> %d, N = %d: CSR launch %.1f us | table launch %.1f us (output stream alone at 5 TB/s: %.1f us); max diff %.1e' % (
        tag, d, cin, cout, N, r[False][0], r[True][0], 4e-6 * N * cout / 5.0, diff), flush=True)
