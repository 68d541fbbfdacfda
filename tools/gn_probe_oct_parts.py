"""Where the time of the sibling-octet GroupNorm launch goes: main rows only / + octet-owned aux rows / + leftovers, by
calling ofx_gn_apply_planes_oct with truncated plans (results of the truncated runs are not meaningful).  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octfusion_amd import _lib, ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_small, split2octree_large
from octfusion_amd._lib import call, ptr, stream
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
which = sys.argv[1] if len(sys.argv) > 1 else 'both'
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
def run(doc, shapes, tag):
    for d, C in shapes:
        N = doc.csr(d)[2]
        x = torch.randn(N, C, device=dev)
        w = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
        mean = torch.zeros(8 * C, device=dev); rstd = torch.ones(8 * C, device=dev)
        bid = doc.batch_id32(d)
        seg_ptr, col, _, _ = doc.csr(d); _, multi_seg, V = doc.ext(d)
        out = torch.empty(N, C, device=dev); aux = torch.empty(V + 1, C, device=dev)
        plan, shift, n_own, n_left, (o_ptr, o_ent, o_head, o_src) = doc.oct_plan(d)
        base = plan.data_ptr()
        n_oct = (N + shift + 7) // 8
        zptr = torch.zeros(n_oct + 1, dtype=torch.int32, device=dev)
        zero_left = torch.zeros(8, dtype=torch.int32, device=dev)
        def go(optr, nown, left, nleft):
            call('ofx_gn_apply_planes_oct', ptr(x), C, N, C, ptr(bid), ptr(mean), ptr(rstd), None, None, 32, 1e-5, 1e-5, ptr(w), ptr(b), 1, 3, ptr(out), C * 4,
                 nown + nleft - 1, ptr(aux), optr, base + 4 * o_ent, nown, shift, left, base + 4 * o_src, nleft, stream())
        t_main = timeit(lambda: go(ptr(zptr), 0, ptr(zero_left), 1))
        t_own = timeit(lambda: go(base + 4 * o_ptr, n_own, ptr(zero_left), 1)) if n_own else float('nan')
        t_left = timeit(lambda: go(ptr(zptr), 0, base + 4 * o_head, n_left))
        t_all = timeit(lambda: go(base + 4 * o_ptr, n_own, base + 4 * o_head, n_left))
        # the same entries with aux row ids renumbered in octet order (sequential writes instead of scattered ones)
        seq = plan.clone()
        seq[o_ent:o_ent + 2 * n_own:2] = torch.arange(1, n_own + 1, dtype=torch.int32, device=dev)
        sbase = seq.data_ptr()
        def go2():
            call('ofx_gn_apply_planes_oct', ptr(x), C, N, C, ptr(bid), ptr(mean), ptr(rstd), None, None, 32, 1e-5, 1e-5, ptr(w), ptr(b), 1, 3, ptr(out), C * 4,
                 n_own, ptr(aux), sbase + 4 * o_ptr, sbase + 4 * o_ent, n_own, shift, ptr(zero_left), sbase + 4 * o_src, 1, stream())
        t_seq = timeit(go2) if n_own else float('nan')
        print('   owned rows with sequential ids: %.1f us' % t_seq)
        call('ofx_set_gn_left_place', 0)
        t_left_i = timeit(lambda: go(ptr(zptr), 0, base + 4 * o_head, n_left))
        t_all_i = timeit(lambda: go(base + 4 * o_ptr, n_own, base + 4 * o_head, n_left))
        call('ofx_set_gn_left_place', 1)
        fo = lambda **kw: None
        print('%s d%d C=%d N=%d: main only %.1f us (%.2f TB/s) | + owned (%d) %.1f | main + leftovers (%d) %.1f | all %.1f || interleaved: main + leftovers %.1f | all %.1f' % (
            tag, d, C, N, t_main, 8e-6 * N * C / t_main, n_own, t_own, n_left, t_left, t_all, t_left_i, t_all_i), flush=True)
oc = split2octree_small(synthetic.shell6_split(8).to(dev), 6, 4)
if which in ('hr', 'both'):
    run(DualOctree(oc), [(6, 128), (6, 256), (6, 384), (5, 256), (5, 768)], 'shell6x8')
if which in ('feature', 'both'):
    x6, y6, z6, _ = oc.xyzb(6)
    run(DualOctree(split2octree_large(oc, synthetic.shell8_split_large(x6, y6, z6), 6)), [(8, 64), (8, 128), (8, 192), (7, 128)], 'shell8x8')
