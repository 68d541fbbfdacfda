"""Per-shape setup cost: octree + dual-graph build (cold = first call incl. lazy kernel loading, warm = repeat)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octfusion_amd import synthetic
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_large, split2octree_small
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
for B in (1, 8):
    split = synthetic.shell6_split(B, jitter=True).to(dev)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        oc = split2octree_small(split, 6, 4)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        doc = DualOctree(oc)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        x6, y6, z6, _ = oc.xyzb(6)
        sl = synthetic.shell8_split_large(x6, y6, z6)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        oc8 = split2octree_large(oc, sl, 6)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        doc8 = DualOctree(oc8)
        torch.cuda.synchronize(); t5 = time.perf_counter()
        print('B=%d rep %d: octree6 %.1f ms, graph6 %.1f ms | octree8 %.1f ms, graph8 %.1f ms (N8 = %d)' % (
            B, rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t4 - t3), 1e3 * (t5 - t4), doc8.total_num))
