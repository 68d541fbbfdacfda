"""The two GroupNorm-apply launches the round-5 review set targets for (depth 6, C = 128 and depth 8, C = 64), each 20
times through the sibling-octet launch with the separate finalize launch -- run under `rocprofv3 --kernel-trace --stats`:
the kernel_stats rows of gn_apply_oct_kernel<3, false> are then per-shape averages (one process per shape).
    python tools/gn_oct_rocprof.py d6|d8"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octfusion_amd import _lib, ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_small, split2octree_large
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
which = sys.argv[1]
ops.GN_OCT_FINALIZE = False
oc = split2octree_small(synthetic.shell6_split(8).to(dev), 6, 4)
if which == 'd8':
    x6, y6, z6, _ = oc.xyzb(6)
    oc = split2octree_large(oc, synthetic.shell8_split_large(x6, y6, z6), 6)
doc = DualOctree(oc)
d, C = (6, 128) if which == 'd6' else (8, 64)
N = doc.csr(d)[2]
gn = M.DualOctreeGroupNorm(C).to(dev)
x = torch.randn(N, C, device=dev)
for _ in range(23):
    gn(x, doc, d, act='silu', planes=ops.planes_mode())
torch.cuda.synchronize()
