"""Tiny workload for PMC runs: a few launches of the fused GraphConv + the same-shape dense GEMM."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octfusion_amd import ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_small
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
oc = split2octree_small(synthetic.shell6_split(8).to(dev), 6, 4)
doc = DualOctree(oc)
N = doc.csr(6)[2]
conv = M.GraphConv(128, 128, 7, 7, 5).to(dev)
x = torch.randn(N, 128, device=dev)
for _ in range(4):
    conv(x, doc, 6)
A = torch.randn(N, 896, device=dev)
pw = ops.PackedWeight().get(torch.randn(896, 128, device=dev), 'kn')
for _ in range(4):
    ops.gemm(A, pw)
torch.cuda.synchronize()
