"""GroupNorm apply with aux rows: sibling-octet launch (round 6) vs block-owned rows (round 4) vs separate aux blocks
(round 3) vs no aux rows, per width, on the shell-6 x 8 (hr) and shell-8 x 8 (feature) trees (GPU box).
    python tools/gn_probe_oct.py [hr|feature|both] [--out FILE]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octfusion_amd import _lib, ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_small, split2octree_large
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
ops.GN_OCT_FINALIZE_MAX_ELEMS = 1 << 40       # (the A/B below decides per shape)
which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 'both'
out = sys.argv[sys.argv.index('--out') + 1] if '--out' in sys.argv else None
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
rows = []
def run(doc, shapes, tag):
    for d, C in shapes:
        N = doc.csr(d)[2]
        gn = M.DualOctreeGroupNorm(C).to(dev)
        x = torch.randn(N, C, device=dev)
        bid, cnt = doc.batch_id32(d), doc.count(d)
        seg_ptr, col, _, _ = doc.csr(d); _, multi_seg, V = doc.ext(d)
        stats = torch.zeros(8 * C * 2, dtype=torch.float64, device=dev)
        _lib.call('ofx_gn_stats', x.data_ptr(), C, N, C, bid.data_ptr(), 8, stats.data_ptr(), torch.cuda.current_stream().cuda_stream)
        f = lambda **kw: ops.group_norm(x, bid, cnt, 8, gn.weights, gn.bias, gn.group, act='silu', stats=stats, **kw)
        op, bp = doc.oct_plan(d), doc.aux_plan(d)
        t_plain = timeit(lambda: f(planes=3))
        t_oct = timeit(lambda: f(planes=3, aux_graph=(seg_ptr, col, multi_seg, V, op)))
        t_blk = timeit(lambda: f(planes=3, aux_graph=(seg_ptr, col, multi_seg, V, bp)))
        t_sep = timeit(lambda: f(planes=3, aux_graph=(seg_ptr, col, multi_seg, V)))
        t_oct2 = timeit(lambda: f(planes=3, aux_graph=(seg_ptr, col, multi_seg, V, op)))
        ops.GN_OCT_FINALIZE = False
        t_oct_sep = timeit(lambda: f(planes=3, aux_graph=(seg_ptr, col, multi_seg, V, op)))
        ops.GN_OCT_FINALIZE = True
        r = dict(tree=tag, depth=d, C=C, N=N, V=V, oct_owned=op[2], oct_left=op[3], block_left=bp[1], us_no_aux=t_plain,
                 us_oct=min(t_oct, t_oct2), us_oct_separate_finalize=t_oct_sep, us_block=t_blk, us_sep=t_sep,
                 TBps_oct=(8e-6 * N * C + 4e-6 * (V + 1) * C) / min(t_oct, t_oct2))
        rows.append(r)
        print('%s d%d C=%d N=%d V=%d (oct-owned %d, left %d | block left %d): no aux %.1f us | oct %.1f (with the finalize launch %.1f) | block %.1f | sep %.1f  -> %.2f TB/s' % (
            tag, d, C, N, V, op[2], op[3], bp[1], t_plain, r['us_oct'], t_oct_sep, t_blk, t_sep, r['TBps_oct']), flush=True)
oc = split2octree_small(synthetic.shell6_split(8).to(dev), 6, 4)
if which in ('hr', 'both'):
    run(DualOctree(oc), [(6, 128), (6, 256), (6, 384), (5, 128), (5, 256), (5, 384), (5, 768), (4, 256), (4, 512), (4, 64)], 'shell6x8')
if which in ('feature', 'both'):
    x6, y6, z6, _ = oc.xyzb(6)
    run(DualOctree(split2octree_large(oc, synthetic.shell8_split_large(x6, y6, z6), 6)), [(8, 64), (8, 128), (8, 192), (7, 128), (7, 192), (7, 384)], 'shell8x8')
if out:
    json.dump(rows, open(out, 'w'), indent=1)
