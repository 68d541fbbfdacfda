import os, sys
sys.path.insert(0, '/root/repo')
import torch
from octfusion_amd import _lib, ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_small, split2octree_large
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
oc = split2octree_small(synthetic.shell6_split(8).to(dev), 6, 4)
x6, y6, z6, _ = oc.xyzb(6)
doc = DualOctree(split2octree_large(oc, synthetic.shell8_split_large(x6, y6, z6), 6))
def timeit(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
for d, C in [(8, 64), (8, 128), (8, 192), (7, 128), (7, 256)]:
    N = doc.csr(d)[2]
    gn = M.DualOctreeGroupNorm(C).to(dev)
    x = torch.randn(N, C, device=dev)
    bid, cnt = doc.batch_id32(d), doc.count(d)
    seg_ptr, col, _, _ = doc.csr(d); _, multi_seg, V = doc.ext(d)
    stats = torch.zeros(8 * C * 2, dtype=torch.float64, device=dev)
    _lib.call('ofx_gn_stats', x.data_ptr(), C, N, C, bid.data_ptr(), 8, stats.data_ptr(), torch.cuda.current_stream().cuda_stream)
    f = lambda **kw: ops.group_norm(x, bid, cnt, 8, gn.weights, gn.bias, gn.group, act='silu', stats=stats, **kw)
    plan = doc.aux_plan(d)
    t0 = timeit(lambda: f()); t3 = timeit(lambda: f(planes=3)); t3p = timeit(lambda: f(planes=3, aux_graph=(seg_ptr, col, multi_seg, V, plan)))
    print('d%d C=%d N=%d V=%d: fp32 %.0f us (%.2f TB/s)  planes %.0f us  planes+aux(plan) %.0f us  (+%.0f%% for %.0f%% more rows)' % (d, C, N, V, t0, 8e-6*N*C/t0, t3, t3p, 100*(t3p/t3-1), 100.0*V/N))
