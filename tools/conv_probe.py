"""Micro-probe of the fused GraphConv kernel on the bench shape (depth 6, C=128): separates the
gather's memory behaviour from the kernel's internal pipeline.  Run on the GPU box."""
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


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for d, cin, cout in [(6, 128, 128), (6, 256, 256), (5, 256, 256), (4, 512, 512)]:
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, d - 1).to(dev)
    x = torch.randn(N, cin, device=dev)
    flops = 2.0 * N * 7 * (cin + d - 1) * cout
    t = timeit(lambda: conv(x, doc, d))
    conv.emit_stats = False
    t_nostats = timeit(lambda: conv(x, doc, d))
    conv.emit_stats = True
    nbr_ext, multi_seg, V = doc.ext(d)
    # variant: every (row, dir) gathers the row itself (perfect locality, same instruction stream)
    self_tab = torch.arange(N, device=dev, dtype=torch.int32).repeat_interleave(7).contiguous()
    saved = doc._ext[d]
    doc._ext[d] = (self_tab, multi_seg, 0)
    t_self = timeit(lambda: conv(x, doc, d))
    # variant: random neighbours (no locality at all)
    rnd = torch.randint(0, N, (N * 7,), device=dev, dtype=torch.int32)
    doc._ext[d] = (rnd, multi_seg, 0)
    t_rnd = timeit(lambda: conv(x, doc, d))
    doc._ext[d] = saved
    # dense GEMM of the same shape
    A = torch.randn(N, 7 * cin, device=dev)
    pw = ops.PackedWeight().get(torch.randn(7 * cin, cout, device=dev), 'kn')
    t_dense = timeit(lambda: ops.gemm(A, pw))
    dyy = torch.randn(N, cout, device=dev)
    t_dx = timeit(lambda: ops.graphconv_backward(x, dyy, doc, d, conv.weights, d - 1, need_dw=False), n=10)
    t_dw = timeit(lambda: ops.graphconv_backward(x, dyy, doc, d, conv.weights, d - 1, need_dx=False), n=10)
    print('   backward: dx %.3f ms (%.0f TF)  dW %.3f ms (%.0f TF)' % (t_dx, 2.0 * N * 7 * cin * cout / t_dx / 1e9, t_dw, flops / t_dw / 1e9))
    print('d%d N=%d cin=%d cout=%d: graph %.3f ms (%.0f TF)  no-stats %.3f ms  self-gather %.3f ms  random-gather %.3f ms  dense %.3f ms (%.0f TF)'
          % (d, N, cin, cout, t, flops / t / 1e9, t_nostats, t_self, t_rnd, t_dense, 2.0 * N * 7 * cin * cout / t_dense / 1e9))
