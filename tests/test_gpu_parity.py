"""GPU parity: the HIP path (through the C ABI) against (a) golden vectors captured from
the reference's own code and (b) the CPU oracle on the same seeded inputs.

Bar: bit-exact for integer / index work (after canonical edge order); fp32 results within
1e-3 (north_star) -- tested tighter (2e-4 per module): the default contraction is bf16x3 on the
bf16 matrix pipe (~1e-5 per layer), exact fp32 MFMA is the ofx_set_precision(1) mode; both are
tested (test_precision_modes_vs_oracle, tests/test_gpu_fullwidth.py).
`close()` measures max |a - b| / max |b| (error relative to the tensor's range), NOT an element-wise
relative error; tests/test_gpu_fullwidth.py reports both figures at the real network widths."""
import pytest
import torch

import common as C

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

REL = 2e-4


def dev():
    return torch.device('cuda:0')


def close(a, b, rel=REL):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-6)
    err = float((a - b).abs().max()) / scale
    assert err < rel, 'rel-to-max error %.3e' % err


def canon(edge_idx, edge_dir):
    row, col = edge_idx[0].cpu(), edge_idx[1].cpu()
    ed = edge_dir.cpu()
    n = int(max(int(row.max()), int(col.max())) + 1)
    order = torch.argsort((row * 7 + ed) * n + col, stable=True)
    return torch.stack([row[order], col[order]]), ed[order]


def check_octree(oc, rec):
    assert oc.depth == rec['depth']
    for d in range(oc.depth + 1):
        assert int(oc.nnum[d]) == int(rec['nnum'][d])
        assert int(oc.nnum_nempty[d]) == int(rec['nnum_nempty'][d])
        assert torch.equal(oc.keys[d].cpu(), rec['keys'][d])
        assert torch.equal(oc.children[d].cpu().long(), rec['children'][d].long())


def check_doctree(doc, rec):
    assert doc.total_num == rec['total_num']
    assert torch.equal(doc.nnum, rec['nnum']) and torch.equal(doc.lnum, rec['lnum'])
    assert torch.equal(doc.ncum, rec['ncum'])
    for d, r in rec['graph'].items():
        g = doc.graph[d]
        assert bool((torch.diff(g['edge_idx'][0] * 7 + g['edge_dir']) >= 0).all())   # sort_edges order
        ei, ed = canon(g['edge_idx'], g['edge_dir'])
        assert ed.numel() == r['E'] and g['node_type'].numel() == r['N']
        assert C.sha_int(ei) == r['sha_edge_idx']
        assert C.sha_int(ed) == r['sha_edge_dir']
        assert C.sha_int(g['node_type'].cpu()) == r['sha_node_type']
        assert C.sha_int(g['keyd'].cpu()) == r['sha_keyd']
        assert C.sha_int(g['node_mask'].cpu()) == r['sha_node_mask']
        assert C.sha_int(doc.batch_id(d).cpu()) == r['sha_batch_id']
        if 'edge_idx' in r:
            assert torch.equal(ei, r['edge_idx'].long())


def build(split, depth, fd):
    from octfusion_amd.octree import split2octree_small
    from octfusion_amd.dual_octree import DualOctree
    oc = split2octree_small(split.to(dev()), depth, fd)
    doc = DualOctree(oc)
    doc.post_processing_for_docnn()
    return oc, doc


def tiny(split):
    return build(split, 4, 2)


def small(split):
    return build(split, 5, 3)


def load(module, keys):
    sd = C.fill_state_dict(keys)
    module.load_state_dict(sd, strict=True)
    return module.to(dev()).eval()


# ------------------------------------------------------------------ integer work
def test_scan():
    from octfusion_amd import ops
    for n in [1, 7, 2048, 2049, 100000, 1 << 21]:
        x = torch.randint(0, 5, (n,), dtype=torch.int32)
        out = ops.scan_i32(x.to(dev())).cpu()
        ref = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(x.long(), 0)])
        assert torch.equal(out.long(), ref)


def test_octree_and_graph_tiny(golden):
    from octfusion_amd.octree import split2octree_large
    from octfusion_amd.dual_octree import DualOctree
    G = golden('g_octree_graph')
    t = G['tiny']
    oc, doc = tiny(t['split_small'])
    check_octree(oc, t['octree'])
    check_doctree(doc, t['doctree'])
    tl = G['tiny_large']
    oc_l = split2octree_large(oc, tl['split_large'].to(dev()), 4)
    check_octree(oc_l, tl['octree'])
    check_doctree(DualOctree(oc_l), tl['doctree'])


def test_octree_and_graph_shell(golden):
    from octfusion_amd.octree import split2octree_large
    from octfusion_amd.dual_octree import DualOctree
    G = golden('g_octree_graph')
    s6 = G['shell6_b2']
    oc, doc = build(C.shell6_split(2, jitter=True), 6, 4)
    assert torch.equal(oc.nnum, s6['octree_nnum'])
    for d in range(7):
        assert C.sha_int(oc.keys[d].cpu()) == s6['sha_keys'][d]
        assert C.sha_int(oc.children[d].cpu()) == s6['sha_children'][d]
    check_doctree(doc, s6['doctree'])
    s8 = G['shell8_b1']
    oc6, _ = build(C.shell6_split(1), 6, 4)
    x, y, z, b = oc6.xyzb(6)
    oc8 = split2octree_large(oc6, C.shell8_split_large(x.cpu(), y.cpu(), z.cpu()).to(dev()), 6)
    assert torch.equal(oc8.nnum, s8['octree_nnum'])
    doc8 = DualOctree(oc8)
    check_doctree(doc8, s8['doctree'])
    assert doc8.csr(8)[2] == 448232 and doc8.csr(8)[3] == 3374048


def test_graph_vs_oracle_random():
    """Random ragged trees (incl. an all-empty batch element) against the CPU oracle."""
    from oracle import dual_octree as OD, sampler as OS
    for seed, B, p in [(21, 1, 0.2), (22, 3, 0.6), (23, 2, 0.05)]:
        split = C.random_split_small(B, 2, seed, p=p)
        if B == 3:
            split[1] = -1.0                       # one element with nothing below the full layer
        oc, doc = tiny(split)
        o_oc = OS.split2octree_small(split, 4, 2)
        o_doc = OD.OracleDualOctree(o_oc)
        o_doc.post_processing_for_docnn()
        for d in range(2, 5):
            a = canon(doc.graph[d]['edge_idx'], doc.graph[d]['edge_dir'])
            b = OD.canonical_edges(o_doc.graph[d]['edge_idx'], o_doc.graph[d]['edge_dir'])
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
            assert torch.equal(doc.graph[d]['node_type'].cpu(), o_doc.graph[d]['node_type'])
            assert torch.equal(doc.graph[d]['keyd'].cpu(), o_doc.graph[d]['keyd'])
            assert torch.equal(doc.graph[d]['node_mask'].cpu(), o_doc.graph[d]['node_mask'])
            assert torch.equal(doc.batch_id(d).cpu(), o_doc.batch_id(d))


def test_octree2voxel_roundtrip():
    from octfusion_amd import ops
    from oracle.octree import Octree, octree2voxel
    for B, d, Cc in [(2, 2, 5), (1, 4, 64), (3, 3, 70)]:
        data = torch.randn(B * 8 ** d, Cc)
        oc = Octree(d, d, B)
        for k in range(d + 1):
            oc.octree_grow_full(k)
        ref = octree2voxel(data, oc, d).permute(0, 4, 1, 2, 3).contiguous()
        vox = ops.octree2voxel_cf(data.to(dev()), B, d)
        assert torch.equal(vox.cpu(), ref)
        back = ops.voxel2octree_cf(vox, d)
        assert torch.equal(back.cpu(), data)


# ------------------------------------------------------------------ fp32 modules vs golden
def test_modules(golden):
    from octfusion_amd import modules as M, ops
    from oracle import modules as OM
    G = golden('g_modules')
    oc, doc = tiny(G['split_small'])
    N4 = doc.csr(4)[2]
    N3 = doc.csr(3)[2]

    # stand-alone gather vs the reference's scatter_mean(x[col], row*7+dir)
    x = C.rand_input('gm', N4, 12)
    g = doc.graph[4]
    idx = (g['edge_idx'][0] * 7 + g['edge_dir']).cpu()
    ref = OM.scatter_mean(x[g['edge_idx'][1].cpu()], idx, N4 * 7)
    got = ops.gather_mean(x.to(dev()), *doc.csr(4)[:2])
    close(got.reshape(N4 * 7, 12), ref, 1e-6)

    for name in ['gc_nt0', 'gc_nt3_bias', 'gc_d3', 'gc_c64']:
        r = G[name]
        cin, cout, et, deg, nt, bias = r['args']
        m = load(M.GraphConv(cin, cout, et, deg, nt, use_bias=bias), r['keys'])
        x = C.rand_input(name, doc.csr(r['d'])[2], cin).to(dev())
        close(m(x, doc, r['d']), r['out'])
    for name in ['gn12', 'gn64', 'gn60', 'gn96']:
        r = G[name]
        m = load(M.DualOctreeGroupNorm(r['c']), r['keys'])
        assert m.group == r['group']
        x = (C.rand_input(name, N4, r['c']) * 2 + 0.5).to(dev())
        close(m(data=x, doctree=doc, depth=4), r['out'])
    m = load(M.Downsample(6), G['down']['keys'])
    close(m(C.rand_input('down', 40, 6).to(dev())), G['down']['out'])
    m = load(M.Upsample(6), G['up']['keys'])
    close(m(C.rand_input('up', 5, 6).to(dev())), G['up']['out'])
    r = G['gdown']
    m = load(M.GraphDownsample(*r['args']), r['keys'])
    close(m(C.rand_input('gdown', N4, 8).to(dev()), doc, 4), r['out'])
    r = G['gup']
    m = load(M.GraphUpsample(*r['args']), r['keys'])
    close(m(C.rand_input('gup', N3, 8).to(dev()), doc, 3), r['out'])
    for name in ['rbe_diff', 'rbe_same']:
        r = G[name]
        m = load(M.GraphResBlockEmbed(*r['args']), r['keys'])
        cin = r['args'][0]
        y = m(C.rand_input(name, N4, cin).to(dev()), C.rand_input(name + 'e', 2, 32).to(dev()), doc, 4)
        close(y, r['out'])
    r = G['resblocks']
    m = load(M.GraphResBlocks(*r['args']), r['keys'])
    close(m(C.rand_input('resblocks', N4, 8).to(dev()), doc, 4), r['out'])
    r = G['c1x1gngelu']
    m = load(M.Conv1x1GnGeluSequential(8, 32), r['keys'])
    close(m((C.rand_input('c1x1gngelu', N4, 8).to(dev()), doc, 4)), r['out'])


def test_graphconv_wide_vs_oracle():
    """MFMA fast path (cin % 32 == 0, several k-tiles and n-tiles, type slab, ragged M)."""
    from octfusion_amd import modules as M
    from oracle import dual_octree as OD, modules as OM, sampler as OS
    split = C.random_split_small(2, 3, 31, p=0.45)
    oc, doc = small(split)
    o_oc = OS.split2octree_small(split, 5, 3)
    o_doc = OD.OracleDualOctree(o_oc)
    o_doc.post_processing_for_docnn()
    for d, cin, cout, nt, bias in [(5, 64, 160, 4, True), (4, 128, 96, 3, False), (5, 32, 3, 4, False),
                                   (3, 96, 130, 0, True)]:
        m = M.GraphConv(cin, cout, 7, 7, nt, use_bias=bias)
        keys = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
        sd = C.fill_state_dict(keys)
        m.load_state_dict(sd)
        m = m.to(dev())
        N = doc.csr(d)[2]
        x = C.rand_input('wide%d' % d, N, cin)
        ref = OM.graph_conv(x, o_doc, d, sd['weights'], sd.get('bias'), nt)
        close(m(x.to(dev()), doc, d), ref)
        # fused epilogue terms
        emb = C.rand_input('wide_e', 2, cout)
        res = C.rand_input('wide_r', N, cout)
        ref2 = ref + emb[o_doc.batch_id(d)] + res
        close(m(x.to(dev()), doc, d, emb=emb.to(dev()), res=res.to(dev())), ref2)


def test_fused_statistics_and_edge_shapes():
    """(a) GroupNorm sums produced by the GraphConv epilogue (two-stage partials + mixed-batch waves + col path for
    Cin % 32 != 0) equal the stand-alone gn_stats pass on the same output, on a ragged batch of 5 whose element
    boundaries fall inside tiles and waves; (b) widths that are not multiples of 4 / 32 and M = 0, 1 rows."""
    from octfusion_amd import modules as M, ops
    from oracle import dual_octree as OD, modules as OM, sampler as OS
    split = C.random_split_small(5, 3, 41, p=0.4)
    split[3] = -1.0                                   # an element with nothing below the full layer
    oc, doc = small(split)
    o_doc = OD.OracleDualOctree(OS.split2octree_small(split, 5, 3))
    o_doc.post_processing_for_docnn()
    for d, cin, cout, nt in [(5, 64, 128, 4), (4, 32, 64, 3), (5, 24, 32, 4), (5, 3, 128, 4)]:
        m = M.GraphConv(cin, cout, 7, 7, nt)
        sd = C.fill_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()])
        m.load_state_dict(sd)
        m = m.to(dev())
        N = doc.csr(d)[2]
        x = C.rand_input('fs%d_%d' % (d, cin), N, cin).to(dev())
        res = C.rand_input('fsr%d_%d' % (d, cin), N, cout).to(dev())
        stats = torch.zeros(5 * cout * 2, dtype=torch.float64, device=dev())
        pw = m._pw.get(m.weights, 'graphconv', cin, nt)
        seg_ptr, col, _, _ = doc.csr(d)
        y = ops.graphconv(x, doc.nbr(d), seg_ptr, col, pw, cin, doc.type_frac(d, nt), None, None, doc.batch_id32(d),
                          res, None, ext=doc.ext(d), stats=stats)
        ref = OM.graph_conv(x.cpu(), o_doc, d, sd['weights'], None, nt) + res.cpu()
        close(y, ref)
        bid = doc.batch_id32(d).long()
        want = torch.zeros(5, cout, 2, dtype=torch.float64, device=dev())
        want[:, :, 0].index_add_(0, bid, y.double())
        want[:, :, 1].index_add_(0, bid, y.double() ** 2)
        got = stats.view(5, cout, 2)
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
        # and the norm that consumes them equals the oracle norm of the same tensor
        gn = M.DualOctreeGroupNorm(cout).to(dev())
        setattr(y, ops.STATS_ATTR, stats)
        wgt, bs = C.rand_input('fsw%d' % cout, 1, cout), C.rand_input('fsb%d' % cout, 1, cout)
        gn.load_state_dict({'weights': wgt, 'bias': bs})
        want_n = OM.dual_octree_group_norm(y.cpu(), o_doc, d, wgt, bs, 32)
        close(gn(y, doc, d), want_n, 1e-4)
    # (b) odd shapes through the dense contraction
    for Mrows, K, Ncol in [(0, 64, 32), (1, 32, 5), (127, 96, 66), (129, 40, 130), (300, 7, 1)]:
        A = torch.randn(Mrows, K)
        W = torch.randn(K, Ncol)
        b = torch.randn(Ncol)
        y = ops.gemm(A.to(dev()), ops.PackedWeight().get(W.to(dev()), 'kn'), bias=b.to(dev()))
        assert tuple(y.shape) == (Mrows, Ncol)
        if Mrows:
            close(y, (A.double() @ W.double() + b.double()).float(), 1e-4)


def test_graphconv_backward_vs_autograd():
    """Training path: dx (fused gather-GEMM over the reverse graph) and dW (TN fp32-MFMA kernel)
    against torch.autograd of the fp64 oracle GraphConv, on fast and generic layer shapes; the reverse graph
    itself against a host construction."""
    from octfusion_amd import ops
    from oracle import dual_octree as OD, modules as OM, sampler as OS
    split = C.random_split_small(3, 3, 51, p=0.45)
    oc, doc = small(split)
    o_doc = OD.OracleDualOctree(OS.split2octree_small(split, 5, 3))
    o_doc.post_processing_for_docnn()
    # reverse graph == transpose of the forward edge list with weights 1 / |forward segment|
    for d in (4, 5):
        g = o_doc.graph[d]
        row, col = g['edge_idx']
        key = row * 7 + g['edge_dir']
        cnt = torch.bincount(key, minlength=int(key.max()) + 1)[key].float()
        rkey = col * 7 + g['edge_dir']
        order = torch.argsort(rkey * (int(row.max()) + 1) + row)
        rv = doc.rev(d)
        n7 = doc.csr(d)[2] * 7
        assert torch.equal(rv['rev_ptr'].cpu().long(),
                           torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(torch.bincount(rkey, minlength=n7), 0)]))
        assert torch.equal(rv['rev_row'].cpu().long(), row[order])
        assert torch.equal(rv['rev_w'].cpu(), (1.0 / cnt)[order])
    for d, cin, cout, nt in [(5, 64, 128, 4), (4, 32, 96, 3), (5, 96, 32, 0), (5, 3, 64, 4), (4, 64, 4, 3)]:
        ntt = nt if nt > 1 else 0
        N = doc.csr(d)[2]
        x = C.rand_input('bx%d_%d' % (d, cin), N, cin)
        dy = C.rand_input('bdy%d_%d' % (d, cout), N, cout)
        W = C.rand_input('bw%d_%d' % (cin, cout), 7 * (cin + ntt), cout) * 0.1
        with torch.enable_grad():
            x64 = x.double().requires_grad_(True)
            W64 = W.double().requires_grad_(True)
            y = OM.graph_conv(x64, o_doc, d, W64, None, nt)
            (y * dy.double()).sum().backward()
        dx, dW = ops.graphconv_backward(x.to(dev()), dy.to(dev()), doc, d, W.to(dev()), nt)
        close(dx, x64.grad.float(), 1e-4)
        close(dW, W64.grad.float(), 5e-5)                      # bf16x3 TN contraction (exact-fp32 mode: 1e-5, below)


def test_group_norm_backward_vs_autograd():
    """DualOctreeGroupNorm (+SiLU / GELU) backward against torch.autograd of the fp64 oracle norm, ragged batch."""
    import torch.nn.functional as F
    from octfusion_amd import ops
    from oracle import dual_octree as OD, modules as OM, sampler as OS
    split = C.random_split_small(4, 3, 52, p=0.4)
    split[2] = -1.0
    oc, doc = small(split)
    o_doc = OD.OracleDualOctree(OS.split2octree_small(split, 5, 3))
    o_doc.post_processing_for_docnn()
    for d, Cc, act in [(5, 64, None), (4, 128, 'silu'), (5, 24, 'gelu'), (5, 96, 'silu')]:
        N = doc.csr(d)[2]
        groups = OM.gn_groups(Cc)
        x = C.rand_input('gbx%d_%d' % (d, Cc), N, Cc) * 1.5 + 0.3
        dy = C.rand_input('gbdy%d_%d' % (d, Cc), N, Cc)
        w = C.rand_input('gbw%d' % Cc, 1, Cc) * 0.5 + 1.0
        b = C.rand_input('gbb%d' % Cc, 1, Cc) * 0.2
        with torch.enable_grad():
            x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
            y = OM.dual_octree_group_norm(x64, o_doc, d, w64, b64, 32)
            if act == 'silu':
                y = F.silu(y)
            elif act == 'gelu':
                y = F.gelu(y)
            (y * dy.double()).sum().backward()
        dx, dg, db = ops.group_norm_backward(x.to(dev()), dy.to(dev()), doc.batch_id32(d), doc.count(d), doc.batch_size,
                                             w.to(dev()), b.to(dev()), groups, act=act)
        close(dx, x64.grad.float(), 2e-5)
        close(dg, w64.grad.float().reshape(-1), 2e-5)
        close(db, b64.grad.float().reshape(-1), 2e-5)


def test_dense_backward_pieces():
    """gemm_tn (P^T Q on the fp32 MFMA, slice-ordered reduce) and the Linear / Conv1x1 backward built on it."""
    from octfusion_amd import ops
    for rows, K, N in [(1000, 64, 128), (33, 132, 4), (5000, 8, 260), (1, 4, 4)]:
        P, Q = torch.randn(rows, K), torch.randn(rows, N)
        close(ops.gemm_tn(P.to(dev()), Q.to(dev())), (P.double().t() @ Q.double()).float(), 5e-5)
        ops.set_precision('fp32')
        try:
            close(ops.gemm_tn(P.to(dev()), Q.to(dev())), (P.double().t() @ Q.double()).float(), 1e-5)
        finally:
            ops.set_precision(ops.DEFAULT_PRECISION)
    x, dy, W = torch.randn(777, 96), torch.randn(777, 40), torch.randn(40, 96)
    dx, dW, db = ops.linear_backward(x.to(dev()), dy.to(dev()), W.to(dev()))
    close(dx, (dy.double() @ W.double()).float(), 1e-4)
    close(dW, (dy.double().t() @ x.double()).float(), 5e-5)
    close(db, dy.sum(0), 1e-5)


def test_resblock_backward_vs_autograd():
    """Whole GraphResBlockEmbed backward (norm+SiLU, GraphConv + time embedding, norm+SiLU, GraphConv, skip 1x1)
    assembled from the gradient kernels, against torch.autograd through the fp64 oracle block."""
    from octfusion_amd import modules as M, backward as BW
    from oracle import dual_octree as OD, modules as OM, sampler as OS
    split = C.random_split_small(3, 3, 53, p=0.45)
    oc, doc = small(split)
    o_doc = OD.OracleDualOctree(OS.split2octree_small(split, 5, 3))
    o_doc.post_processing_for_docnn()
    for d, cin, cout in [(5, 64, 96), (4, 64, 64)]:
        blk = M.GraphResBlockEmbed(cin, 48, 0.0, cout, 7, 7, d - 1)
        sd = C.fill_state_dict([(k, tuple(v.shape)) for k, v in blk.state_dict().items()])
        sd['conv2.weights'] = C.rand_input('rb_c2_%d' % cout, *sd['conv2.weights'].shape) * 0.05    # not the zero init
        blk.load_state_dict(sd)
        blk = blk.to(dev())
        N = doc.csr(d)[2]
        x = C.rand_input('rbx%d' % d, N, cin)
        emb = C.rand_input('rbe%d' % d, 3, 48)
        dy = C.rand_input('rbdy%d' % d, N, cout)
        with torch.enable_grad():
            sd64 = {k: v.double().requires_grad_(True) for k, v in sd.items()}
            x64, e64 = x.double().requires_grad_(True), emb.double().requires_grad_(True)
            y = OM.graph_resblock_embed(x64, e64, o_doc, d, sd64, d - 1)
            (y * dy.double()).sum().backward()
        close(blk(x.to(dev()), emb.to(dev()), doc, d), y.detach().float(), 1e-4)
        dx, demb, grads = BW.graph_resblock_embed_backward(blk, x.to(dev()), emb.to(dev()), doc, d, dy.to(dev()))
        close(dx, x64.grad.float(), 2e-4)
        close(demb, e64.grad.float(), 2e-4)
        assert set(grads) == set(sd)
        for k, gval in grads.items():
            close(gval, sd64[k].grad.float(), 2e-4)


def to_rows(vox, depth):
    from octfusion_amd import ops
    return ops.voxel2octree_cf(vox.to(dev()).contiguous(), depth)


def to_vox(rows, B, depth):
    from octfusion_amd import ops
    return ops.octree2voxel_cf(rows, B, depth)


def test_gridconv_vs_torch():
    """27-tap gather-GEMM conv (stride 1 / stride 2 / upsample+conv), fast + slow + split-K paths."""
    import torch.nn.functional as F
    from octfusion_amd import graph_unet_lr as LR
    for B, d, cin, cout, mode in [(2, 3, 64, 48, 0), (2, 3, 16, 64, 0), (1, 4, 32, 8, 0), (2, 3, 64, 64, 1),
                                  (2, 2, 96, 160, 2), (1, 2, 256, 128, 0), (3, 1, 64, 32, 0)]:
        S = 1 << d
        m = LR.GridConv3d(cin, cout, mode)
        keys = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
        sd = C.fill_state_dict(keys)
        m.load_state_dict(sd)
        m = m.to(dev())
        x = C.rand_input('gridconv%d%d' % (cin, mode), B, cin, S, S, S)
        if mode == 0:
            ref = F.conv3d(x, sd['weight'], sd['bias'], padding=1)
        elif mode == 1:
            ref = F.conv3d(x, sd['weight'], sd['bias'], stride=2, padding=1)
        else:
            ref = F.conv3d(F.interpolate(x, scale_factor=2, mode='nearest'), sd['weight'], sd['bias'], padding=1)
        gs = LR.GridState(B, d, dev())
        y = m(to_rows(x, d), gs)
        close(to_vox(y, B, m.out_depth(d)), ref)


def test_gridconv_backward_vs_autograd():
    """27-tap conv backward (dx over the reverse tap table, dW on the TN kernel) against torch.autograd of
    F.conv3d in fp64: stride 1, stride 2, nearest-upsample + conv; fast and generic channel counts."""
    import torch.nn.functional as F
    from octfusion_amd import graph_unet_lr as LR, backward as BW
    for B, d, cin, cout, mode in [(2, 3, 64, 32, 0), (2, 3, 16, 64, 0), (2, 3, 64, 64, 1), (2, 2, 32, 96, 2),
                                  (1, 2, 64, 8, 2), (3, 1, 64, 32, 0)]:
        S = 1 << d
        m = LR.GridConv3d(cin, cout, mode)
        sd = C.fill_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()])
        m.load_state_dict(sd)
        m = m.to(dev())
        x = C.rand_input('gcb%d%d' % (cin, mode), B, cin, S, S, S)
        d_out = m.out_depth(d)
        So = 1 << d_out
        dy = C.rand_input('gcbdy%d%d' % (cout, mode), B, cout, So, So, So)
        with torch.enable_grad():
            x64 = x.double().requires_grad_(True)
            w64 = sd['weight'].double().requires_grad_(True)
            b64 = sd['bias'].double().requires_grad_(True)
            if mode == 0:
                y = F.conv3d(x64, w64, b64, padding=1)
            elif mode == 1:
                y = F.conv3d(x64, w64, b64, stride=2, padding=1)
            else:
                y = F.conv3d(F.interpolate(x64, scale_factor=2, mode='nearest'), w64, b64, padding=1)
            (y * dy.double()).sum().backward()
        gs = LR.GridState(B, d, dev())
        dx, dw, db = BW.gridconv_backward(m, to_rows(x, d), to_rows(dy, d_out), gs)
        close(to_vox(dx, B, d), x64.grad.float(), 1e-4)
        close(dw, w64.grad.float(), 5e-5)
        close(db, b64.grad.float(), 1e-5)


def test_attention_backward_vs_autograd():
    """QKVAttention backward (dq, dk, dv) against torch.autograd of the reference formula in fp64; the forward of
    the same inputs is checked on the way."""
    import math
    from octfusion_amd import ops
    for B, T, heads, ch in [(2, 64, 4, 16), (1, 512, 4, 32), (3, 8, 2, 64), (2, 100, 1, 24)]:
        Cc = heads * ch
        qkv = C.rand_input('attb%d_%d' % (T, ch), B * T, 3 * Cc)
        dout = C.rand_input('attbd%d_%d' % (T, ch), B * T, Cc)
        with torch.enable_grad():
            x = qkv.double().requires_grad_(True)
            t = x.view(B, T, heads, 3, ch).permute(0, 2, 3, 4, 1)          # [B, heads, 3, ch, T]
            q, k, v = t[:, :, 0], t[:, :, 1], t[:, :, 2]
            scale = 1 / math.sqrt(math.sqrt(ch))
            w = torch.softmax(torch.einsum('bhct,bhcs->bhts', q * scale, k * scale), dim=-1)
            o = torch.einsum('bhts,bhcs->bhct', w, v)                        # [B, heads, ch, T]
            out = o.permute(0, 3, 1, 2).reshape(B * T, Cc)
            (out * dout.double()).sum().backward()
        close(ops.attention(qkv.to(dev()), B, T, heads), out.detach().float(), 1e-4)
        close(ops.attention_backward(qkv.to(dev()), dout.to(dev()), B, T, heads), x.grad.float(), 1e-4)


def test_lr_unet_backward_vs_autograd(golden):
    """Whole dense lr U-Net (time embedding, ResnetBlocks, attention, stride-2 / upsample convs, skip concats):
    forward + backward assembled from the gradient kernels against torch.autograd through the oracle net --
    every one of the state_dict's parameters and the input."""
    from octfusion_amd import graph_unet_lr as LR, backward as BW, ops
    from oracle import unet as OU
    keys = golden('g_dense')['lr']['keys']
    cfg = C.TINY_LR_CFG
    B, S = 2, 8
    sd = C.fill_state_dict(keys)
    net = load(LR.UNet3DModel(**cfg), keys)
    x = C.rand_input('lrb_x', B, 8, S, S, S)
    xsc = C.rand_input('lrb_sc', B, 8, S, S, S)
    t = torch.tensor([0.3, -1.2])
    dyv = C.rand_input('lrb_dy', B, 8, S, S, S)
    with torch.enable_grad():
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xin = torch.cat([x, xsc], 1).requires_grad_(True)
        y = OU.lr_forward(sdg, dict(cfg, num_classes=None), xin[:, :8], t, xin[:, 8:], None)
        (y * dyv).sum().backward()
    rows = ops.voxel2octree_cf(torch.cat([x, xsc], 1).to(dev()), 3)
    dy_rows = ops.voxel2octree_cf(dyv.to(dev()), 3)
    yr, dx, grads = BW.lr_unet_forward_backward(net, rows, B, t.to(dev()), lambda yy: dy_rows)
    close(ops.octree2voxel_cf(yr, B, 3), y.detach(), 1e-3)
    close(ops.octree2voxel_cf(dx, B, 3), xin.grad, 5e-3)      # fp32 autograd reference through ~40 layers
    assert set(grads) == set(sd), set(sd) ^ set(grads)
    gmax = max(float(v.grad.abs().max()) for v in sdg.values())
    for k in sd:
        # biases / time projections that feed a GroupNorm with one channel per group (C <= 32) have an exactly
        # zero gradient: both sides are then fp32 summation noise, so the tolerance has an absolute floor tied to
        # the largest gradient of the net
        ref = sdg[k].grad
        err = float((grads[k].cpu() - ref).abs().max())
        assert err <= 5e-3 * float(ref.abs().max()) + 2e-5 * gmax, (k, err, float(ref.abs().max()), gmax)


def test_hr_unet_backward_vs_autograd(golden):
    """The sparse hr U-Net with the nested lr net (input conv, res blocks, graph down / up-sampling with pooled
    GEMMs and leaf copies, skip concats, middle blocks around the dense net, end norm, output conv): forward +
    backward assembled from the gradient kernels against torch.autograd through the oracle -- every parameter of
    both nets and the input."""
    from octfusion_amd import graph_unet_union as U, backward as BW
    from oracle import dual_octree as OD, modules as OM, sampler as OS, unet as OU
    G = golden('g_unet')
    r = G['uncond']
    oc, doc = small(G['split_small'])
    o_doc = OD.OracleDualOctree(OS.split2octree_small(G['split_small'], 5, 3))
    o_doc.post_processing_for_docnn()
    sd = C.fill_state_dict(r['keys'])
    # the zero-initialised modules get real weights so that every gradient path carries signal
    net = load(U.UNet3DModel(**union_cfg(None)), r['keys'])
    N = doc.total_num
    x = C.rand_input('hrb_x', N, 3)
    dy = C.rand_input('hrb_dy', N, 3)
    t = torch.tensor([0.4, -0.9])
    with torch.enable_grad():
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xg = x.clone().requires_grad_(True)
        parts = {p: OM._sub(sdg, p) for p in ('unet_lr', 'unet_hr')}
        y = OU.hr_forward(parts['unet_hr'], C.TINY_HR_CFG, xg, o_doc, t, None, parts['unet_lr'], C.TINY_LR_CFG)
        (y * dy).sum().backward()
    yy, dx, g_hr, g_lr = BW.hr_unet_forward_backward(net.unet_hr, x.to(dev()), doc, net.unet_lr, t.to(dev()),
                                                     lambda out: dy.to(dev()))
    close(yy, y.detach(), 1e-3)
    close(dx, xg.grad, 5e-3)
    grads = {'unet_hr.' + k: v for k, v in g_hr.items()}
    grads.update({'unet_lr.' + k: v for k, v in g_lr.items()})
    # the lr net runs as the middle: its input / output convolutions take no part
    used = {k for k in sd if sdg[k].grad is not None}
    assert set(grads) == used, (set(grads) ^ used)
    gmax = max(float(sdg[k].grad.abs().max()) for k in used)
    for k in used:
        ref = sdg[k].grad
        err = float((grads[k].cpu() - ref).abs().max())
        assert err <= 5e-3 * float(ref.abs().max()) + 2e-5 * gmax, (k, err, float(ref.abs().max()), gmax)


def test_activation_checkpointing_gives_the_same_gradients(golden):
    """ldm_diffusion_util.py:158-169 / modules.py:730-743 (use_checkpoint): a checkpointed res block keeps only its
    input and recomputes GroupNorm -> conv1 -> GroupNorm in the backward -- same kernels on the same inputs, so the
    output, the input gradient and every parameter gradient are BIT-identical to the plain path; and the module's own
    flag (the constructor argument of the reference) is what switches it."""
    from octfusion_amd import graph_unet_union as U, backward as BW
    G = golden('g_unet')
    r = G['uncond']
    oc, doc = small(G['split_small'])
    net = load(U.UNet3DModel(**union_cfg(None)), r['keys'])
    N = doc.total_num
    x = C.rand_input('hrb_x', N, 3)
    dy = C.rand_input('hrb_dy', N, 3)
    t = torch.tensor([0.4, -0.9])

    def run():
        return BW.hr_unet_forward_backward(net.unet_hr, x.to(dev()), doc, net.unet_lr, t.to(dev()), lambda out: dy.to(dev()))
    kept = []
    orig = BW._gres_fwd

    def spy(*a, **k):
        y, saved = orig(*a, **k)
        kept.append(sum(v is not None for v in saved))
        return y, saved
    BW._gres_fwd = spy
    try:
        y0, dx0, g0, gl0 = run()
        assert set(kept) == {4}
        del kept[:]
        blocks = [m for m in net.unet_hr.modules() if hasattr(m, 'use_checkpoint') and hasattr(m, 'block1_norm')]
        assert blocks and not any(b.use_checkpoint for b in blocks)
        for b in blocks:
            b.use_checkpoint = True                      # what UNet3DModel(use_checkpoint=True) sets (graph_unet_hr.py:66)
        y1, dx1, g1, gl1 = run()
        assert set(kept) == {1}                          # only the block input is kept
    finally:
        BW._gres_fwd = orig
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    assert set(g0) == set(g1) and all(torch.equal(g0[k], g1[k]) for k in g0)
    assert all(torch.equal(gl0[k], gl1[k]) for k in gl0)


def test_feature_unet_backward_vs_autograd(golden):
    """3-stage cascade: the feature net with the hr net nested as its middle (itself running as_middle, without an
    lr net) -- forward + backward against autograd through the oracle, all parameters that take part."""
    from octfusion_amd import graph_unet_union as U, backward as BW
    from octfusion_amd.octree import split2octree_large
    from octfusion_amd.dual_octree import DualOctree
    from oracle import dual_octree as OD, modules as OM, sampler as OS, unet as OU
    G = golden('g_unet')
    r = G['feature']
    oc, _ = small(G['split_small'])
    doc_l = DualOctree(split2octree_large(oc, r['split_large'].to(dev()), 5))
    o_doc = OD.OracleDualOctree(OS.split2octree_large(OS.split2octree_small(G['split_small'], 5, 3), r['split_large'], 5))
    o_doc.post_processing_for_docnn()
    cfg = r['cfg']
    sd = C.fill_state_dict(r['keys'])
    net = load(U.UNet3DModel(**cfg), r['keys'])
    N = doc_l.total_num
    x = C.rand_input('ftb_x', N, 3)
    dy = C.rand_input('ftb_dy', N, 3)
    t = torch.tensor([-0.4, 0.7])

    feat = dict(input_depth=7, full_depth=3, model_channels=cfg['model_channels'][2],
                channel_mult=cfg['channel_mult'][2], num_res_blocks=cfg['num_res_blocks'][2], num_classes=None)
    mid = dict(kind='hr', input_depth=5, full_depth=3, model_channels=cfg['model_channels'][1],
               channel_mult=cfg['channel_mult'][1], num_res_blocks=cfg['num_res_blocks'][1], num_classes=None)
    with torch.enable_grad():
        sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xg = x.clone().requires_grad_(True)
        y = OU.hr_forward(OM._sub(sdg, 'unet_feature'), feat, xg, o_doc, t, None, OM._sub(sdg, 'unet_hr'), mid)
        (y * dy).sum().backward()
    yy, dx, g_ft, g_hr = BW.hr_unet_forward_backward(net.unet_feature, x.to(dev()), doc_l, net.unet_hr, t.to(dev()),
                                                     lambda out: dy.to(dev()))
    close(yy, y.detach(), 1e-3)
    close(dx, xg.grad, 5e-3)
    # and one optimisation step of the feature stage runs (x0 objective, Objaverse-style)
    from octfusion_amd import training as TR
    opt = TR.AdamW(dict(net.named_parameters()), lr=1e-3)
    l0 = TR.hr_stage_step(net, opt, x.to(dev()), doc_l, 7, stage='feature', df_type='x0')
    assert l0 == l0 and l0 > 0
    net.load_state_dict(sd)                               # restore for the gradient comparison below
    grads = {'unet_feature.' + k: v for k, v in g_ft.items()}
    grads.update({'unet_hr.' + k: v for k, v in g_hr.items()})
    used = {k for k in sd if sdg[k].grad is not None}
    assert set(grads) == used, (set(grads) ^ used)
    gmax = max(float(sdg[k].grad.abs().max()) for k in used)
    for k in used:
        ref = sdg[k].grad
        err = float((grads[k].cpu() - ref).abs().max())
        assert err <= 5e-3 * float(ref.abs().max()) + 2e-5 * gmax, (k, err, float(ref.abs().max()), gmax)


def test_hr_training_step(golden):
    """Second-stage training step (eps objective) on the union net: first-step loss equals the oracle's, and the
    loss falls on a fixed batch."""
    from octfusion_amd import graph_unet_union as U, training as TR
    from oracle import dual_octree as OD, modules as OM, sampler as OS, unet as OU
    G = golden('g_unet')
    r = G['uncond']
    oc, doc = small(G['split_small'])
    o_doc = OD.OracleDualOctree(OS.split2octree_small(G['split_small'], 5, 3))
    o_doc.post_processing_for_docnn()
    sd = C.fill_state_dict(r['keys'])
    net = load(U.UNet3DModel(**union_cfg(None)), r['keys'])
    opt = TR.AdamW(dict(net.named_parameters()), lr=1e-3)
    N = doc.total_num
    codes = C.rand_input('hrt_codes', N, 3)
    noise = C.rand_input('hrt_noise', N, 3)
    times = torch.tensor([0.35, 0.8])
    ls = OS.beta_linear_log_snr(times)
    a, sg = OS.log_snr_to_alpha_sigma(ls)
    bid = o_doc.batch_id(5)
    noised = a[bid].unsqueeze(1) * codes + sg[bid].unsqueeze(1) * noise
    parts = {p: OM._sub(sd, p) for p in ('unet_lr', 'unet_hr')}
    ref = OU.hr_forward(parts['unet_hr'], C.TINY_HR_CFG, noised, o_doc, ls.float(), None, parts['unet_lr'], C.TINY_LR_CFG)
    ref_loss = float(torch.nn.functional.mse_loss(ref, noise))
    losses = [TR.hr_stage_step(net, opt, codes.to(dev()), doc, 5, times.to(dev()), noise.to(dev())) for _ in range(6)]
    assert abs(losses[0] - ref_loss) <= 1e-3 * ref_loss, (losses[0], ref_loss)
    assert losses[-1] < 0.9 * losses[0], losses


def test_adamw_ema_and_lr_training_step(golden):
    """ofx_adamw_step / ofx_ema_update against torch.optim.AdamW + the reference's EMA rule over several steps on the
    same gradients; then the lr-stage training step (noise -> predict x0 -> MSE -> backward -> AdamW -> EMA) lowers
    the loss on a fixed batch and matches a torch-autograd training run of the oracle net step for step."""
    import copy
    from octfusion_amd import graph_unet_lr as LR, training as TR
    from oracle import unet as OU, sampler as OS
    # --- optimiser kernels
    torch.manual_seed(3)
    p0 = torch.randn(1000)
    ref = torch.nn.Parameter(p0.clone())
    topt = torch.optim.AdamW([ref], lr=3e-3)
    mine = torch.nn.Parameter(p0.clone().to(dev()))
    opt = TR.AdamW({'w': mine}, lr=3e-3)
    ema_ref, ema_mine = p0.clone(), torch.nn.Parameter(p0.clone().to(dev()))

    class _M(torch.nn.Module):
        def __init__(self, p):
            super().__init__()
            self.w = p
    for it in range(5):
        g = torch.randn(1000)
        ref.grad = g.clone()
        topt.step()
        opt.step({'w': g.to(dev())})
        ema_ref = ema_ref * 0.99 + (1 - 0.99) * ref.data
        TR.ema_update(_M(ema_mine), _M(mine), 0.99)
    torch.testing.assert_close(mine.data.cpu(), ref.data, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ema_mine.data.cpu(), ema_ref, rtol=1e-5, atol=1e-6)
    # --- lr stage
    keys = golden('g_dense')['lr']['keys']
    cfg = C.TINY_LR_CFG
    sd = C.fill_state_dict(keys)
    net = load(LR.UNet3DModel(**cfg), keys)
    ema = copy.deepcopy(net)
    opt = TR.AdamW(dict(net.named_parameters()), lr=2e-3)
    B, S = 2, 8
    split = C.random_split_small(B, 3, 9, p=0.4)
    g = torch.Generator().manual_seed(11)
    times = [torch.rand(B, generator=g) for _ in range(6)]
    noises = [torch.randn(B, 8, S, S, S, generator=g) for _ in range(6)]
    # torch reference: the oracle net under autograd + torch.optim.AdamW
    with torch.enable_grad():
        rp = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ropt = torch.optim.AdamW(list(rp.values()), lr=2e-3)
        ref_losses = []
        for t, nz in zip(times, noises):
            ls = OS.beta_linear_log_snr(t)
            a, sg = OS.log_snr_to_alpha_sigma(ls)
            noised = a.view(B, 1, 1, 1, 1) * split + sg.view(B, 1, 1, 1, 1) * nz
            out = OU.lr_forward(rp, dict(cfg, num_classes=None), noised, ls.float(), None, None)
            loss = torch.nn.functional.mse_loss(out, split)
            ropt.zero_grad()
            loss.backward()
            ropt.step()
            ref_losses.append(float(loss.detach()))
    losses = [TR.lr_stage_step(net, opt, split.to(dev()), t.to(dev()), nz.to(dev()), ema=ema, ema_rate=0.9)
              for t, nz in zip(times, noises)]
    # identical forward at step 0; afterwards Adam's sign-like first updates amplify gradient noise on the
    # exactly-zero-gradient parameters, so the two runs only track each other
    assert abs(losses[0] - ref_losses[0]) <= 1e-4 * ref_losses[0], (losses, ref_losses)
    assert abs(losses[1] - ref_losses[1]) <= 1e-2 * ref_losses[1], (losses, ref_losses)
    assert all(abs(a - b) <= 0.1 * b for a, b in zip(losses, ref_losses)), (losses, ref_losses)
    assert losses[-1] < 0.7 * losses[0] and ref_losses[-1] < 0.7 * ref_losses[0]
    # same batch, same noise: the loss goes down
    fixed = [TR.lr_stage_step(net, opt, split.to(dev()), times[0].to(dev()), noises[0].to(dev())) for _ in range(8)]
    assert fixed[-1] < 0.9 * fixed[0], fixed
    assert all(bool(torch.isfinite(p).all()) for p in ema.parameters())


def test_precision_modes_vs_oracle():
    """Every contraction mode against the fp64 oracle on one 128 -> 128 GraphConv: exact fp32 ~2e-6, the default
    fp16x3 (fp16 hi + lo pairs) in the same class, bf16x3 (bf16 pairs) ~1e-5, both through the planes kernel and the
    register-staged one."""
    from octfusion_amd import modules as M, ops
    from oracle import dual_octree as OD, modules as OM, sampler as OS
    split = C.random_split_small(2, 3, 33, p=0.45)
    oc, doc = small(split)
    o_oc = OS.split2octree_small(split, 5, 3)
    o_doc = OD.OracleDualOctree(o_oc)
    o_doc.post_processing_for_docnn()
    m = M.GraphConv(128, 128, 7, 7, 4)
    sd = C.fill_state_dict([(k, tuple(v.shape)) for k, v in m.state_dict().items()])
    m.load_state_dict(sd)
    m = m.to(dev())
    x = C.rand_input('prec', doc.csr(5)[2], 128)
    ref = OM.graph_conv(x.double(), o_doc, 5, sd['weights'].double(), None, 4).float()
    scale = float(ref.abs().max())
    assert ops.get_precision() == ops.DEFAULT_PRECISION == 'fp16x3'
    errs = {}
    saved = ops.PLANES_MIN_TILES
    try:
        for prec, planes in (('fp32', False), ('fp16x3', True), ('fp16x3', False), ('bf16x3', True), ('bf16x3', False)):
            ops.set_precision(prec)
            ops.PLANES_MIN_TILES = 1 if planes else (1 << 30)
            y = m(x.to(dev()), doc, 5, split_input=True).cpu()
            errs[(prec, planes)] = float((y - ref).abs().max()) / scale
    finally:
        ops.set_precision(ops.DEFAULT_PRECISION)
        ops.PLANES_MIN_TILES = saved
    assert errs[('fp32', False)] < 5e-6, errs
    assert errs[('fp16x3', True)] < 5e-6 and errs[('fp16x3', False)] < 5e-6, errs
    assert errs[('bf16x3', True)] < 5e-5 and errs[('bf16x3', False)] < 5e-5, errs
    assert errs[('fp16x3', True)] < 0.3 * errs[('bf16x3', True)], errs        # the point of the fp16 pairs


def test_dense_blocks_reference_signatures(golden):
    """ResnetBlock.forward(x, time_emb), AttentionBlock.forward(x), ConvDownsample / ConvUpsample.forward(x)
    on [b, c, D, H, W] tensors (modules.py:63-95, 474-547) -- called exactly the way the reference calls them, against
    the reference's own outputs (g_dense: attn / attn512 / resnet; g_boundary: convdown / convup / resnet_same)."""
    from octfusion_amd import graph_unet_lr as LR
    G, Gb = golden('g_dense'), golden('g_boundary')
    m = load(LR.AttentionBlock(32, num_heads=4), G['attn']['keys'])
    y = m(G['attn']['x'].to(dev()))
    assert y.shape == G['attn']['out'].shape
    close(y, G['attn']['out'])
    m = load(LR.AttentionBlock(128, num_heads=4), G['attn512']['keys'])
    close(m(G['attn512']['x'].to(dev())), G['attn512']['out'])
    m = load(LR.ResnetBlock(3, 8, 12, emb_dim=16, dropout=0.0), G['resnet']['keys'])
    close(m(G['resnet']['x'].to(dev()), G['resnet']['emb'].to(dev())), G['resnet']['out'])
    m = load(LR.ResnetBlock(3, 32, 32, emb_dim=16, dropout=0.0), Gb['resnet_same']['keys'])
    close(m(Gb['resnet_same']['x'].to(dev()), Gb['resnet_same']['emb'].to(dev())), Gb['resnet_same']['out'])
    m = load(LR.ConvDownsample(16, dims=3), Gb['convdown']['keys'])
    y = m(Gb['convdown']['x'].to(dev()))
    assert tuple(y.shape) == (2, 16, 4, 4, 4)
    close(y, Gb['convdown']['out'])
    m = load(LR.ConvUpsample(16, dims=3), Gb['convup']['keys'])
    y = m(Gb['convup']['x'].to(dev()))
    assert tuple(y.shape) == (2, 16, 8, 8, 8)
    close(y, Gb['convup']['out'])


def test_dense_and_unet(golden):
    from octfusion_amd import graph_unet_lr as LR, graph_unet_union as U, ops
    G = golden('g_dense')
    r = G['attn']
    m = load(LR.AttentionBlock(32, num_heads=4), r['keys'])
    close(to_vox(m(to_rows(r['x'], 2), LR.GridState(2, 2, dev())), 2, 2), r['out'])
    r = G['attn512']
    m = load(LR.AttentionBlock(128, num_heads=4), r['keys'])
    close(to_vox(m(to_rows(r['x'], 3), LR.GridState(1, 3, dev())), 1, 3), r['out'])
    r = G['resnet']
    m = load(LR.ResnetBlock(3, 8, 12, emb_dim=16, dropout=0.0), r['keys'])
    y = m(to_rows(r['x'], 2), ops.act(r['emb'].to(dev()), 'silu'), LR.GridState(2, 2, dev()))
    close(to_vox(y, 2, 2), r['out'])
    r = G['lr']
    lr = load(LR.UNet3DModel(**C.TINY_LR_CFG), r['keys'])
    close(lr(x=r['x'].to(dev()), timesteps=r['t'].to(dev()), x_self_cond=r['x_self_cond'].to(dev())), r['out'], 1e-3)
    close(lr(x=r['x'].to(dev()), timesteps=r['t'].to(dev())), r['out_nosc'], 1e-3)
    oc, doc = small(G['split_small'])
    m = G['lr_mid']
    close(lr.forward_as_middle(m['h'].to(dev()), doc, m['t'].to(dev()), None, None), m['out'], 1e-3)

    G = golden('g_unet')
    oc, doc = small(G['split_small'])
    for name in ['uncond', 'cond']:
        r = G[name]
        cfg = union_cfg(r['num_classes'])
        net = load(U.UNet3DModel(**cfg), r['keys'])
        label = r['label'].to(dev()) if r['label'] is not None else None
        y = net(unet_type='hr', x=r['x'].to(dev()), doctree=doc, unet_lr=net.unet_lr,
                timesteps=r['t'].to(dev()), x_self_cond=None, label=label)
        close(y, r['out'], 1e-3)
    from octfusion_amd.octree import split2octree_large
    from octfusion_amd.dual_octree import DualOctree
    r = G['feature']
    oc_l = split2octree_large(oc, r['split_large'].to(dev()), 5)
    doc_l = DualOctree(oc_l)
    net = load(U.UNet3DModel(**r['cfg']), r['keys'])
    y = net(unet_type='feature', x=r['x'].to(dev()), doctree=doc_l, unet_lr=net.unet_hr,
            timesteps=r['t'].to(dev()), x_self_cond=None, label=None)
    close(y, r['out'], 1e-3)


def union_cfg(num_classes=None):
    h, l = C.TINY_HR_CFG, C.TINY_LR_CFG
    cfg = dict(stage_flag='hr', image_size=[8, 32], input_depth=[3, 5], unet_type=['lr', 'hr'],
               full_depth=3, input_channels=[8, 3], out_channels=[8, 3],
               model_channels=[l['model_channels'], h['model_channels']],
               num_res_blocks=[[1, 1, 1], h['num_res_blocks']], attention_resolutions=[2, 4],
               channel_mult=[l['channel_mult'], h['channel_mult']], num_heads=4,
               use_checkpoint=False, dims=3)
    if num_classes is not None:
        cfg['num_classes'] = num_classes
    return cfg


def test_vae_decoder(golden):
    """GraphVAE.decode_code on libofx: fixed octree vs golden; growth path (argmax -> split -> grow ->
    rebuilt dual graph on device) vs golden counts + logits."""
    from octfusion_amd.graph_vae import GraphVAE
    from octfusion_amd.octree import split2octree_large
    from octfusion_amd.dual_octree import DualOctree
    G = golden('g_vae')
    oc, doc = tiny(G['split_small'])
    oc_l = split2octree_large(oc, G['split_large'].to(dev()), 4)
    doc_l = DualOctree(oc_l)
    vae = load(GraphVAE(**G['cfg']), G['keys'])
    code = C.rand_input('vae_code', doc_l.csr(4)[2], 3).to(dev())
    out = vae.decode_code(code, doc_l, update_octree=False)
    for d in (4, 5, 6):
        close(out['logits'][d], G['logits'][d], 1e-3)
        close(out['reg_voxs'][d], G['reg_voxs'][d], 1e-3)
    out2 = vae.decode_code(code, DualOctree(split2octree_large(oc, G['split_large'].to(dev()), 4)), update_octree=True)
    # decisions are argmax over 2 logits: identical unless a margin is below the fp tolerance
    for d in (4, 5, 6):
        ref = G['grow']['logits'][d]
        margin = (ref[:, 0] - ref[:, 1]).abs().min()
        if d == 4 or float(margin) > 1e-3 * float(ref.abs().max()):
            close(out2['logits'][d], ref, 1e-3)
        assert tuple(out2['reg_voxs'][d].shape) == G['grow']['reg_shapes'][d]
    assert torch.equal(out2['octree_out'].nnum[:7], G['grow']['nnum'])


@pytest.mark.gpu
def test_neural_mpu_golden(golden):
    """ofx_mpu_eval against the reference's own NeuralMPU outputs (g_mpu.pt): mask bit-exact, sdf to 1e-4."""
    from octfusion_amd.mpu import NeuralMPU
    from octfusion_amd.octree import split2octree_large
    G = golden('g_mpu')
    oc, _ = tiny(G['split_small'])
    oc_l = split2octree_large(oc, G['split_large'].to(dev()), 4)
    fd, ds, dp = G['cfg']
    ncum = torch.cumsum(oc_l.nnum, 0)
    reg = {d: C.rand_input('mpu_code_%d' % d, int(ncum[d] - (ncum[fd - 1] if fd else 0)), 4).to(dev())
           for d in range(ds, dp + 1)}
    out = NeuralMPU(fd, ds, dp)(G['pos'].to(dev()), reg, oc_l)
    for d in range(ds, dp + 1):
        assert torch.equal(out[d][1].cpu(), G['mask'][d])
        torch.testing.assert_close(out[d][0].cpu(), G['sdf'][d], rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_neural_mpu_shell_vs_oracle_and_grid():
    """Shell-6 tree, batch 2: random + lattice-aligned + out-of-cube points against oracle/mpu.py; then the
    in-kernel lattice sweep (calc_sdf) must equal evaluating the same lattice as explicit points, bit for bit."""
    from oracle import mpu as OMPU
    from oracle import sampler as OS
    from octfusion_amd import mpu as M
    from octfusion_amd.octree import split2octree_small
    split = C.shell6_split(2, jitter=True)
    oc = split2octree_small(split.to(dev()), 6, 4)
    oc_o = OS.split2octree_small(split, 6, 4)
    assert torch.equal(oc.nnum[:7], oc_o.nnum[:7])
    fd, dp = 4, 6
    rows = int(oc.nnum[fd:dp + 1].sum())
    reg = C.rand_input('mpu_shell_code', rows, 4)
    g = torch.Generator().manual_seed(5)
    n = 20000
    pos = torch.cat([torch.rand(n, 3, generator=g) * 2.2 - 1.1, torch.randint(0, 2, (n, 1), generator=g).float()], 1)
    pos[:2000, :3] = torch.randint(0, 129, (2000, 3), generator=g).float() / 64 - 1
    sdf, mask = M.mpu_eval(oc, fd, dp, pos.to(dev()), reg.to(dev()))
    want, wmask = OMPU.linear_pred(pos, oc_o, reg, fd, dp)
    assert torch.equal(mask.cpu(), wmask)
    torch.testing.assert_close(sdf.cpu(), want, rtol=1e-4, atol=1e-5)
    assert int(mask.sum()) > 100                      # the shell is actually hit
    # lattice sweep == explicit points
    field = M.MpuField(fd, dp, reg.to(dev()), oc)
    size = 48
    grid = M.calc_sdf(field, batch_size=2, size=size, bbmin=-0.9, bbmax=0.9)
    generic = M.calc_sdf(lambda p: field(p), batch_size=2, size=size, max_batch=30000, bbmin=-0.9, bbmax=0.9)
    assert grid.shape == (2, size, size, size)
    assert torch.equal(grid, generic)
    # and the lattice coordinates follow numpy's fp32 arithmetic (util_dualoctree.py:102-103)
    import numpy as np
    ax = np.arange(size, dtype=np.float32) * ((0.9 - -0.9) / size) + -0.9
    pts = torch.tensor(np.stack(np.meshgrid(ax, ax, ax, indexing='ij'), -1).reshape(-1, 3))
    pts = torch.cat([pts, torch.ones(pts.shape[0], 1)], 1)
    assert torch.equal(field(pts.to(dev())), grid[1].reshape(-1))


def test_split_formats_roundtrip():
    """octree2split_small / _large on device == oracle, and octree -> splits -> octree is the identity
    (util_dualoctree.py:199-273); also through the on-disk sample format (gen_split.py:50-54)."""
    import tempfile
    from oracle import sampler as OS
    from octfusion_amd import checkpoint as CK
    from octfusion_amd.octree import (split2octree_small, split2octree_large, octree2split_small,
                                      octree2split_large)
    split = C.shell6_split(1, jitter=True)
    oc6 = split2octree_small(split.to(dev()), 6, 4)
    sl = C.random_split_large(int(oc6.nnum[6]), 3, p=0.4)
    oc8 = split2octree_large(oc6, sl.to(dev()), 6)
    o8 = OS.split2octree_large(OS.split2octree_small(split, 6, 4), sl, 6)
    ss, sL = octree2split_small(oc8, 4), octree2split_large(oc8, 6)
    assert torch.equal(ss.cpu(), OS.octree2split_small(o8, 4))
    assert torch.equal(sL.cpu(), OS.octree2split_large(o8, 6))
    with tempfile.TemporaryDirectory() as tmp:
        CK.write_splits(tmp, oc8)
        a, b = CK.read_splits(tmp, dev())
    back = split2octree_large(split2octree_small(a, 6, 4), b, 6)
    for d in range(9):
        assert torch.equal(back.keys[d], oc8.keys[d])
        if d < 8:
            assert torch.equal(back.children[d], oc8.children[d])


def test_checkpoint_layout_roundtrip(golden):
    """df_<label>.pth layout (octfusion_model_union.py:501-545): a file written with the reference's keys loads
    strict=True into a fresh model and reproduces the golden output."""
    import tempfile
    import os
    from octfusion_amd import checkpoint as CK
    from octfusion_amd import graph_unet_union as U
    G = golden('g_unet')
    oc, doc = small(G['split_small'])
    r = G['uncond']
    cfg = union_cfg(r['num_classes'])
    sd = C.fill_state_dict(r['keys'])
    lr = {k[len('unet_lr.'):]: v for k, v in sd.items() if k.startswith('unet_lr.')}
    hr = {k[len('unet_hr.'):]: v for k, v in sd.items() if k.startswith('unet_hr.')}
    assert lr and hr and len(lr) + len(hr) == len(sd)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'df_steps-latest.pth')
        torch.save({'df_unet_lr': lr, 'ema_df_unet_lr': lr, 'df_unet_hr': hr, 'ema_df_unet_hr': hr,
                    'opt': {}, 'global_step': 1234}, path)
        net = U.UNet3DModel(**cfg).to(dev()).eval()
        assert CK.load_ckpt(path, net, net) == 1234
        path2 = os.path.join(tmp, 'df_steps-2.pth')
        CK.save_ckpt(path2, net, net, 7)
        again = torch.load(path2, map_location='cpu', weights_only=False)
    assert set(again) == {'df_unet_lr', 'ema_df_unet_lr', 'df_unet_hr', 'ema_df_unet_hr', 'opt', 'global_step'}
    assert list(again['df_unet_hr']) == list(hr) and all(torch.equal(again['df_unet_hr'][k].cpu(), hr[k]) for k in hr)
    y = net(unet_type='hr', x=r['x'].to(dev()), doctree=doc, unet_lr=net.unet_lr, timesteps=r['t'].to(dev()),
            x_self_cond=None, label=None)
    close(y, r['out'], 1e-3)


def test_vae_encoder(golden):
    """GraphVAE encoder (conv1 -> res-blocks -> VAE GraphDownsample x2 -> norm+GELU -> KL_conv) vs the reference
    outputs of g_vae_enc.pt; then encode() -> decode_code() runs end to end."""
    from octfusion_amd.graph_vae import GraphVAE
    from octfusion_amd.octree import split2octree_large
    from octfusion_amd.dual_octree import DualOctree
    G = golden('g_vae_enc')
    oc, _ = tiny(G['split_small'])
    doc_l = DualOctree(split2octree_large(oc, G['split_large'].to(dev()), 4))
    vae = load(GraphVAE(**G['cfg']), G['keys'])
    data = C.rand_input('vae_enc_in', doc_l.csr(6)[2], 4).to(dev())
    convs = vae.octree_encoder_step(data, doc_l)
    close(convs[4][::8], G['h_rows8'], 1e-3)
    code, mean, logvar = vae.encode(data, doc_l, sample=False)
    close(torch.cat([mean, logvar], 1), G['kl'].clamp(min=-1e30), 1e-3)
    noise = torch.randn_like(mean)
    code2, _, _ = vae.encode(data, doc_l, noise=noise)
    torch.testing.assert_close(code2, mean + torch.exp(0.5 * logvar) * noise)
    out = vae.decode_code(code, doc_l, update_octree=False)
    assert set(out['logits']) == {4, 5, 6} and all(bool(torch.isfinite(v).all()) for v in out['reg_voxs'].values())


def _fake_net(shape, device):
    A = torch.linspace(-0.5, 0.5, shape[1]).to(device)

    def net(unet_type=None, x=None, doctree=None, timesteps=None, unet_lr=None, x_self_cond=None, label=None):
        a = A.view(1, -1, *([1] * (x.ndim - 2)))
        tt = timesteps.view(-1, *([1] * (x.ndim - 1))) if x.ndim > 2 else timesteps[0]
        y = torch.tanh(x * 0.7 + a) * 0.9 + 0.05 * torch.tanh(tt)
        if x_self_cond is not None:
            y = y + 0.1 * x_self_cond
        return y
    return net


@pytest.mark.parametrize('name,unet_type,df_type', [('x0', 'lr', 'x0'), ('eps', 'hr', 'eps'), ('x0_graph', 'hr', 'x0')])
def test_sample_loop(golden, name, unet_type, df_type):
    """DDIM driver (libofx update kernels) against the reference's sample_loop outputs; the noise
    the reference drew from torch's CPU RNG is regenerated in the same order and passed explicitly."""
    from octfusion_amd import sampler
    r = golden('g_sample_loop')[name]
    shape = tuple(r['shape'])
    torch.manual_seed(r['seed'])
    init = torch.randn(shape)
    steps = [torch.randn(shape) for _ in range(r['steps'])] if df_type == 'x0' else None
    y = sampler.sample_loop(_fake_net(shape, dev()), shape, r['B'], r['steps'], unet_type, df_type, dev(),
                            truncated_index=r.get('trunc', 0.0), init_noise=init, step_noise=steps)
    close(y, r['out'], 2e-5)


def test_sample_loop_hipgraph_replay(golden):
    """use_graph=True (one eager step, then hipGraph replay per regime) gives the eager result: bit-equal for
    the pure-kernel fake net on the x0 branch (sign + noise + self-conditioning regimes), 1e-5 for a real
    union net on the eps branch (fp64 atomics in the fused statistics may reorder)."""
    from octfusion_amd import sampler, graph_unet_union as U
    r = golden('g_sample_loop')['x0']
    shape = tuple(r['shape'])
    torch.manual_seed(r['seed'])
    init = torch.randn(shape)
    steps = [torch.randn(shape) for _ in range(r['steps'])]
    kw = dict(truncated_index=r.get('trunc', 0.0), init_noise=init, step_noise=steps)
    a = sampler.sample_loop(_fake_net(shape, dev()), shape, r['B'], r['steps'], 'lr', 'x0', dev(), **kw)
    b = sampler.sample_loop(_fake_net(shape, dev()), shape, r['B'], r['steps'], 'lr', 'x0', dev(), use_graph=True, **kw)
    assert torch.equal(a, b)
    close(b, r['out'], 2e-5)
    G = golden('g_unet')
    oc, doc = small(G['split_small'])
    net = load(U.UNet3DModel(**union_cfg(None)), G['uncond']['keys'])
    shp = (doc.total_num, 3)
    init = torch.randn(shp)
    ya = sampler.sample_loop(net, shp, doc.batch_size, 6, 'hr', 'eps', dev(), doctree=doc, unet_lr=net.unet_lr,
                             init_noise=init, use_graph=False)
    yb = sampler.sample_loop(net, shp, doc.batch_size, 6, 'hr', 'eps', dev(), doctree=doc, unet_lr=net.unet_lr,
                             init_noise=init, use_graph=True)
    close(yb, ya, 1e-5)


def test_cascade_two_stage_runs():
    """lr -> octree -> hr -> VAE decode, end to end on device with shrunken nets (shape / sanity checks;
    stage-wise numerics are pinned by the other tests)."""
    from octfusion_amd import configs, graph_unet_union as U, pipeline, synthetic
    from octfusion_amd.graph_vae import GraphVAE
    cfg = dict(configs.SNET_UNCOND, model_channels=[32, 32])
    net = U.UNet3DModel(**{k: v for k, v in dict(cfg, stage_flag='hr').items() if k != 'df_type'})
    net.load_state_dict(synthetic.random_state_dict(net))
    net = net.to(dev()).eval()
    vae = GraphVAE(depth=8, channel_in=4, nout=4, full_depth=4, depth_stop=6, depth_out=8, resblk_type='basic',
                   resblk_num=2, code_channel=16, embed_dim=3)
    vae.load_state_dict(synthetic.random_state_dict(vae))
    vae = vae.to(dev()).eval()
    cs = pipeline.CascadeSampler(net, cfg, vae)
    out = cs.sample(1, ddim_steps=3, split_small=synthetic.shell6_split(1, jitter=False).to(dev()), sdf_resolution=64)
    assert out['octree_small'].nnum.tolist() == [1, 8, 64, 512, 4096, 4672, 20032]
    # SDF sweep of the decoded field (get_sdfs): lattice result == the field evaluated at explicit points
    sdfs = out['sdfs']
    assert tuple(sdfs.shape) == (1, 64, 64, 64) and bool(torch.isfinite(sdfs).all())
    q = torch.tensor([[5, 17, 40], [63, 0, 31]])
    pts = torch.cat([q.float() * (1.8 / 64) - 0.9, torch.zeros(2, 1)], 1).to(dev())
    want = out['decoded']['neural_mpu'](pts)
    torch.testing.assert_close(sdfs[0, q[:, 0], q[:, 1], q[:, 2]], want, rtol=1e-6, atol=1e-7)
    assert tuple(out['hr'].shape) == (25712, 3) and bool(torch.isfinite(out['hr']).all())
    dec = out['decoded']
    assert set(dec['logits']) == {6, 7, 8} and dec['octree_out'].depth == 8
    assert all(bool(torch.isfinite(v).all()) for v in dec['reg_voxs'].values())


def test_full_size_shell6_b8_properties():
    """BASELINE-size input (shell-6, batch 8: N6 = 217 008, E6 = 1 629 600): one fused GraphConv and one
    DualOctreeGroupNorm against the CPU oracle, plus size-independent properties (linearity of the
    convolution; zero mean / unit variance per (batch element, group) after the norm)."""
    from octfusion_amd import modules as M, synthetic
    from oracle import dual_octree as OD, modules as OM, sampler as OS
    split = synthetic.shell6_split(8, jitter=True)
    oc, doc = build(split, 6, 4)
    assert doc.csr(6)[2] == 217008 and doc.csr(6)[3] == 1629600
    o_doc = OD.OracleDualOctree(OS.split2octree_small(split, 6, 4))
    o_doc.post_processing_for_docnn()
    a = canon(doc.graph[6]['edge_idx'], doc.graph[6]['edge_dir'])
    b = OD.canonical_edges(o_doc.graph[6]['edge_idx'], o_doc.graph[6]['edge_dir'])
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])            # bit-exact neighbour indices
    N = doc.csr(6)[2]
    conv = M.GraphConv(128, 128, 7, 7, 5)
    sd = C.fill_state_dict([(k, tuple(v.shape)) for k, v in conv.state_dict().items()])
    conv.load_state_dict(sd)
    conv = conv.to(dev())
    x = C.rand_input('full', N, 128)
    y = conv(x.to(dev()), doc, 6)
    close(y, OM.graph_conv(x, o_doc, 6, sd['weights'], None, 5), 1e-4)
    # linearity: conv(2x + z) - type term == 2 conv(x) + conv(z) - 2 * type term  <=>  check on differences
    z = C.rand_input('full_z', N, 128).to(dev())
    zero = conv(torch.zeros_like(z), doc, 6)                               # = node-type contribution only
    lhs = conv(2 * x.to(dev()) + z, doc, 6) - zero
    rhs = 2 * (y - zero) + (conv(z, doc, 6) - zero)
    close(lhs, rhs, 1e-4)
    gn = M.DualOctreeGroupNorm(128).to(dev())
    g = gn(y, doc, 6)
    bid = doc.batch_id(6)
    for bidx in (0, 3, 7):
        rows = g[bid == bidx].view(-1, 32, 4)
        assert float(rows.mean(dim=(0, 2)).abs().max()) < 1e-4
        assert float((rows.var(dim=(0, 2), unbiased=False) - 1).abs().max()) < 1e-3
    gw = OM.dual_octree_group_norm(y.cpu(), o_doc, 6, torch.ones(1, 128), torch.zeros(1, 128))
    close(g, gw, 1e-4)


@pytest.mark.gpu
def test_incremental_dual_octree(golden):
    """DualOctree(octree, prev=...) (the VAE growth path, graph_vae.py:203-210) adopts every depth whose inputs are
    unchanged and builds only the new one; the result equals a full rebuild, array for array."""
    from octfusion_amd.dual_octree import DualOctree
    from octfusion_amd.octree import split2octree_small
    oc = split2octree_small(C.shell6_split(2, jitter=True).to(dev()), 6, 4)      # depth 6, children[6] all -1
    doc6 = DualOctree(oc)
    doc6.type_frac(6, 5), doc6.pool_maps(6), doc6.unpool_maps(5), doc6.pad_rows(6), doc6.rev(5)      # populate caches
    g = torch.Generator().manual_seed(4)
    label = (torch.rand(int(oc.nnum[6]), generator=g) < 0.4).to(torch.int32).to(dev())
    oc.octree_split(label, 6)
    oc.octree_grow(7)
    oc.depth += 1
    inc = DualOctree(oc, prev=doc6)
    full = DualOctree(oc)
    assert inc.adopted_depths == [4, 5, 6] and full.adopted_depths == []
    assert inc.total_num == full.total_num
    for d in range(4, 8):
        a, b = inc.csr(d), full.csr(d)
        assert a[2:] == b[2:] and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert torch.equal(inc.ext(d)[0], full.ext(d)[0]) and inc.ext(d)[2] == full.ext(d)[2]
        assert torch.equal(inc.batch_id(d), full.batch_id(d)) and torch.equal(inc.count(d), full.count(d))
        for k in ('node_type', 'keyd', 'node_mask', 'edge_idx', 'edge_dir'):
            assert torch.equal(inc.graph[d][k], full.graph[d][k]), (d, k)
        assert torch.equal(inc.type_frac(d, d - 1), full.type_frac(d, d - 1))
        assert torch.equal(inc.pad_rows(d), full.pad_rows(d))
        if d > 4:
            for x, y in zip(inc.pool_maps(d), full.pool_maps(d)):
                assert x == y if isinstance(x, int) else torch.equal(x, y)
        if d < 7:
            for x, y in zip(inc.unpool_maps(d), full.unpool_maps(d)):       # depth 6 changed: must NOT be adopted
                assert x == y if isinstance(x, int) else torch.equal(x, y)
    assert inc.csr(6)[0] is doc6.csr(6)[0]                                   # adopted, not copied or rebuilt
    # a split at the last depth without growth: nothing to build at all
    label7 = (torch.rand(int(oc.nnum[7]), generator=g) < 0.5).to(torch.int32).to(dev())
    oc.octree_split(label7, 7)
    last = DualOctree(oc, prev=inc)
    assert last.adopted_depths == [4, 5, 6, 7] and torch.equal(last.node_child(7), oc.children[7])
    # an unrelated octree adopts nothing
    other = split2octree_small(C.shell6_split(2, jitter=True).to(dev()), 6, 4)
    assert DualOctree(other, prev=doc6).adopted_depths == []


@pytest.mark.gpu
def test_gn_fused_rows_mappings():
    """One-launch GroupNorm of the dense grids: the 16-channel block mapping (C = 64 .. 512) equals the one-block-per-group
    mapping and torch.nn.functional.group_norm; odd widths (cpg 6 / 12: concatenated inputs) and short grids stay on the
    per-group kernel; strided input / output (column slices of wider buffers)."""
    import torch.nn.functional as F
    from octfusion_amd import _lib, ops
    g = torch.Generator().manual_seed(9)
    for B, rows, C in [(3, 4096, 64), (2, 4096, 128), (2, 2048, 256), (1, 2304, 512), (2, 4096, 192), (2, 512, 256), (2, 64, 256), (1, 1000, 64)]:
        x_wide = torch.randn(B * rows, C + 32, generator=g).to(dev())
        x = x_wide[:, 16:16 + C]                                  # strided rows, 64-B aligned start
        w = torch.randn(C, generator=g).to(dev())
        b = torch.randn(C, generator=g).to(dev())
        bid = torch.arange(B).repeat_interleave(rows).to(torch.int32).to(dev())
        cnt = torch.full((B,), float(rows), device=dev())
        outs = {}
        for knob in (1, 0):
            _lib.call('ofx_set_gn_rows16', knob)
            out_wide = torch.zeros(B * rows, C + 8, device=dev())
            y = ops.group_norm(x, bid, cnt, B, w, b, 32, eps=1e-5, act='silu', out=out_wide[:, 4:4 + C], count_eps=0.0,
                               rows_per_batch=rows)
            outs[knob] = y.clone()
            assert float(out_wide[:, :4].abs().max()) == 0.0 and float(out_wide[:, 4 + C:].abs().max()) == 0.0
        _lib.call('ofx_set_gn_rows16', 1)
        ref = F.silu(F.group_norm(x.view(B, rows, C).transpose(1, 2).contiguous().cpu().double(), 32, w.cpu().double(),
                                  b.cpu().double(), 1e-5)).transpose(1, 2).reshape(B * rows, C).float()
        close(outs[1], ref, 2e-5)
        close(outs[0], ref, 2e-5)
        close(outs[1], outs[0], 2e-6)
