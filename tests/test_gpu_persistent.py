"""The persistent stream-K launch of the planes GraphConv (csrc/ofx_gemm3.hip) against the one-tile-per-block launch
(csrc/ofx_gemm2.hip) and the fp64 oracle (modules.py:194-220).

What is specific to this launch shape and therefore checked here:
  * tiles cut by a share boundary are combined from pieces computed by different blocks (in-launch hand-off through
    the workspace + flag words): every output element against the oracle, on layers whose tiles ARE cut (the layer
    shapes below are chosen so that units / blocks is not a multiple of the k tiles per tile);
  * determinism: the same launch twice gives the same bits (pieces are added in ascending k order, never by arrival);
  * the flag words are zero after every launch and the error word never rises;
  * both block geometries, 128- and 64-column tiles, two column tiles per row tile (the table is reused, the weight
    columns change), fused epilogue terms + statistics, fp16 single-pass mode, a layer with fewer tiles than blocks
    (several blocks per tile) and a batch of one shape.
"""
import pytest
import torch

import functools

import common as C
import oracle_cache as OC
from test_gpu_fullwidth import dev, errors, report, shell6, shell6_gpu, shell6_oracle

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _gn_conv_case(prefix, B, d, cin, cout):
    """fp64 oracle of SiLU(GroupNorm(x)) -> GraphConv + emb[batch] + res on the jittered shell-6 batch of B shapes
    (modules.py:194-220, 291-314) -- CPU only."""
    from oracle import modules as OM
    _, o_doc = shell6_oracle(B)
    N = int(o_doc.graph[d]['keyd'].shape[0])
    sd = C.fill_state_dict([('c.weights', (7 * (cin + d - 1), cout)), ('g.weights', (1, cin)), ('g.bias', (1, cin))])
    x = C.rand_input('%s_x_%d_%d' % (prefix, d, cin), N, cin)
    emb = C.rand_input('%s_e_%d' % (prefix, cout), B, cout)
    res = C.rand_input('%s_r_%d_%d' % (prefix, d, cout), N, cout)
    h_ref = OM.silu(OM.dual_octree_group_norm(x.double(), o_doc, d, sd['g.weights'].double(), sd['g.bias'].double()))
    return {'ref': OM.graph_conv(h_ref, o_doc, d, sd['c.weights'].double(), None, d - 1) + emb.double()[o_doc.batch_id(d)]
            + res.double()}


def _conv_case(B, d, cin, cout):
    from oracle import modules as OM
    _, o_doc = shell6_oracle(B)
    N = int(o_doc.graph[d]['keyd'].shape[0])
    w = C.fill_state_dict([('weights', (7 * (cin + d - 1), cout))])['weights']
    x = C.rand_input('pk1_shell6_b1_%d_%d' % (d, cin), N, cin)
    return {'ref': OM.graph_conv(x.double(), o_doc, d, w.double(), None, d - 1)}


PK2_SHAPES = [(6, 128, 128), (5, 256, 256), (6, 64, 64), (6, 384, 128), (4, 128, 128), (5, 128, 320)]
PK8_SHAPES = [(6, 128, 128), (5, 256, 512)]
PK1_SHAPES = [(6, 128, 128), (5, 256, 256), (6, 32, 128)]
for _s in PK2_SHAPES:
    OC.register('pk_%d_%d_%d' % _s, functools.partial(_gn_conv_case, 'pk', 2, *_s))
for _s in PK8_SHAPES:
    OC.register('pk8_%d_%d_%d' % _s, functools.partial(_gn_conv_case, 'pk8', 8, *_s))
for _s in PK1_SHAPES:
    OC.register('pk1_%d_%d_%d' % _s, functools.partial(_conv_case, 1, *_s))


def _run(conv, gn, x, doc, d, mode, emb, res, stats_n):
    from octfusion_amd import ops
    hp = gn(x, doc, d, act='silu', planes=mode)
    with ops.stats_scope(dev()):
        stats = ops.stats_zeros(stats_n, dev()) if stats_n else None
        nt = conv.n_node_type if conv.n_node_type > 1 else 0
        pw2 = conv._pw2.get(conv.weights, conv.in_channels, nt, mode)
        seg_ptr, col, _, _ = doc.csr(d)
        y = ops.graphconv_planes(hp, mode, seg_ptr, col, doc.ext(d), pw2, conv.in_channels, nt,
                                 doc.type_frac_planes(d, nt, mode) if nt else None, None, emb, doc.batch_id32(d), res,
                                 None, stats=stats)
        return y.clone(), (stats.clone() if stats is not None else None)


@pytest.mark.parametrize('prec', ['fp16x3', 'bf16x3', 'fp16'])
def test_persistent_vs_tile_launch_and_oracle(prec):
    from octfusion_amd import _lib, modules as M, ops
    mode = {'fp16x3': 3, 'bf16x3': 2, 'fp16': 1}[prec]
    tol = {'fp16x3': 2e-5, 'bf16x3': 2e-4, 'fp16': 5e-3}[prec]
    B = 2
    oc, doc = shell6_gpu(B)
    saved = ops.get_precision()
    ops.set_precision(prec)
    try:
        # (depth, cin, cout): 128-column tiles with one / two column tiles, 64-column tiles, a wide-K layer, a depth
        # with fewer 256-row tiles than CUs (several blocks share a tile)
        for d, cin, cout in PK2_SHAPES:
            if mode == 1 and cin % 64:
                continue
            nt = d - 1
            conv = M.GraphConv(cin, cout, 7, 7, nt)
            gn = M.DualOctreeGroupNorm(cin)
            sd = C.fill_state_dict([('c.' + k, tuple(v.shape)) for k, v in conv.state_dict().items()] +
                                   [('g.' + k, tuple(v.shape)) for k, v in gn.state_dict().items()])
            conv.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith('c.')})
            gn.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith('g.')})
            conv, gn = conv.to(dev()), gn.to(dev())
            N = doc.csr(d)[2]
            x = C.rand_input('pk_x_%d_%d' % (d, cin), N, cin)
            emb = C.rand_input('pk_e_%d' % cout, B, cout)
            res = C.rand_input('pk_r_%d_%d' % (d, cout), N, cout)
            ref = OC.get('pk_%d_%d_%d' % (d, cin, cout))['ref']
            xg, eg, rg = x.to(dev()), emb.to(dev()), res.to(dev())
            sn = B * cout * 2
            _lib.call('ofx_set_gconv_persistent', 0)
            _lib.call('ofx_set_gconv2_tile', 0)
            y_tile, st_tile = _run(conv, gn, xg, doc, d, mode, eg, rg, sn)
            assert errors(y_tile, ref)['rel_to_max'] < tol
            _lib.call('ofx_set_gconv_persistent', 1)
            for tile in (2, 4, 0):
                _lib.call('ofx_set_gconv2_tile', tile)
                y1, st1 = _run(conv, gn, xg, doc, d, mode, eg, rg, sn)
                y2, st2 = _run(conv, gn, xg, doc, d, mode, eg, rg, sn)
                torch.cuda.synchronize()
                assert not ops.sync_error(dev()), 'a flag wait gave up / flags left set'
                e = errors(y1, ref)
                assert e['rel_to_max'] < tol, (d, cin, cout, tile, e)
                assert torch.equal(y1, y2), 'persistent launch is not deterministic'
                # (statistics of waves that span two batch elements go through fp64 atomics: order-dependent last bits)
                assert float((st1 - st2).abs().max()) <= 1e-12 * float(st1.abs().max())
                # same arithmetic, different summation grouping on cut tiles only
                cross = float((y1 - y_tile).abs().max() / y_tile.abs().max())
                assert cross < (1e-5 if mode == 1 else 2e-6), (d, cin, cout, tile, cross)
                bid = doc.batch_id32(d).long()
                want = torch.zeros(B, cout, 2, dtype=torch.float64, device=dev())
                want[:, :, 0].index_add_(0, bid, y1.double())
                want[:, :, 1].index_add_(0, bid, y1.double() ** 2)
                assert float((st1.view(B, cout, 2) - want).abs().max()) <= 1e-5 * float(want.abs().max())
                report(dict(test='persistent', precision=prec, depth=d, N=N, cin=cin, cout=cout, tile=tile,
                            vs_tile_launch=cross, **e))
    finally:
        _lib.call('ofx_set_gconv2_tile', 0)
        _lib.call('ofx_set_gconv_persistent', 1)
        ops.set_precision(saved)


def test_persistent_single_shape_and_ragged():
    """B = 1 (the generate regime: every depth has fewer tiles than the chip has block slots, so every tile is shared
    by several blocks) and a ragged random batch whose last row tile is partial."""
    from octfusion_amd import _lib, modules as M, ops
    from octfusion_amd.dual_octree import DualOctree
    from octfusion_amd.octree import split2octree_small
    from oracle import dual_octree as OD, modules as OM, sampler as OS
    cases = []
    oc, doc = shell6_gpu(1)
    cases.append(('shell6_b1', doc, None, 1, PK1_SHAPES))
    split = C.random_split_small(3, 3, 77, p=0.45)
    doc_r = DualOctree(split2octree_small(split.to(dev()), 5, 3))
    o_r = OD.OracleDualOctree(OS.split2octree_small(split, 5, 3))
    o_r.post_processing_for_docnn()
    cases.append(('ragged_b3', doc_r, o_r, 3, [(5, 64, 128), (4, 96, 200), (5, 160, 72)]))
    _lib.call('ofx_set_gconv_persistent', 1)
    try:
        for name, dc, oc_, B, shapes in cases:
            for d, cin, cout in shapes:
                nt = d - 1
                conv = M.GraphConv(cin, cout, 7, 7, nt)
                sd = C.fill_state_dict([(k, tuple(v.shape)) for k, v in conv.state_dict().items()])
                conv.load_state_dict(sd)
                conv = conv.to(dev())
                N = dc.csr(d)[2]
                x = C.rand_input('pk1_%s_%d_%d' % (name, d, cin), N, cin)
                ref = (OC.get('pk1_%d_%d_%d' % (d, cin, cout))['ref'] if oc_ is None else
                       OM.graph_conv(x.double(), oc_, d, sd['weights'].double(), None, nt))
                for tile in (2, 4):
                    _lib.call('ofx_set_gconv2_tile', tile)
                    saved = ops.PLANES_MIN_TILES
                    ops.PLANES_MIN_TILES = 1
                    try:
                        y = conv(x.to(dev()), dc, d, split_input=True)
                    finally:
                        ops.PLANES_MIN_TILES = saved
                    torch.cuda.synchronize()
                    assert not ops.sync_error(dev())
                    e = errors(y, ref)
                    assert e['rel_to_max'] < 2e-4, (name, d, cin, cout, tile, e)
                    report(dict(test='persistent_small', case=name, depth=d, N=N, cin=cin, cout=cout, tile=tile, **e))
    finally:
        _lib.call('ofx_set_gconv2_tile', 0)


def test_whole_tile_rounds_plus_region_at_bench_size():
    """The launch shape of the bench workload itself (B = 8: 848 row tiles >= 2 x the block slots, so the launch has
    whole-tile rounds in front of the stream-K region): persistent (rounds + region), persistent (pure stream-K) and
    one-tile-per-block launches of the same layer agree, and the default one matches the fp64 oracle.  One layer with a
    single column tile (halo sharing only) and one with four (the column tiles of a row tile run on one XCD together)."""
    from octfusion_amd import _lib, modules as M, ops
    B = 8
    oc, doc = shell6_gpu(B)
    try:
        for d, cin, cout in PK8_SHAPES:
            nt = d - 1
            conv = M.GraphConv(cin, cout, 7, 7, nt)
            gn = M.DualOctreeGroupNorm(cin)
            sd = C.fill_state_dict([('c.' + k, tuple(v.shape)) for k, v in conv.state_dict().items()] +
                                   [('g.' + k, tuple(v.shape)) for k, v in gn.state_dict().items()])
            conv.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith('c.')})
            gn.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith('g.')})
            conv, gn = conv.to(dev()), gn.to(dev())
            N = doc.csr(d)[2]
            x = C.rand_input('pk8_x_%d_%d' % (d, cin), N, cin)
            emb = C.rand_input('pk8_e_%d' % cout, B, cout)
            res = C.rand_input('pk8_r_%d_%d' % (d, cout), N, cout)
            ref = OC.get('pk8_%d_%d_%d' % (d, cin, cout))['ref']
            xg, eg, rg = x.to(dev()), emb.to(dev()), res.to(dev())
            ys = {}
            for pers in (0, 1, 2):
                _lib.call('ofx_set_gconv_persistent', pers)
                ys[pers], _ = _run(conv, gn, xg, doc, d, ops.planes_mode(), eg, rg, B * cout * 2)
                torch.cuda.synchronize()
                assert not ops.sync_error(dev())
            e = errors(ys[1], ref)
            assert e['rel_to_max'] < 2e-5, (d, cin, cout, e)
            for pers in (1, 2):
                cross = float((ys[pers] - ys[0]).abs().max() / ys[0].abs().max())
                assert cross < 2e-6, (d, cin, cout, pers, cross)
            report(dict(test='persistent_b8', depth=d, N=N, cin=cin, cout=cout, **e))
    finally:
        _lib.call('ofx_set_gconv_persistent', 1)


@pytest.mark.parametrize('persistent', [1, 0])
def test_epilogue_batch_labels_inside_one_wave(persistent):
    """The three epilogue paths of the planes GraphConv by the number of batch elements inside one 64-row wave:
    one (per-wave partial sums -> second-stage reduce), two (two sets of sums, one atomic pair per column and
    element), three or more (per-lane runs) -- an element with fewer than 64 nodes at a depth, which no synthetic tree of
    the suite has.  The kernel only reads the labels, so they are planted: a real depth-6 layer, its time-embedding add
    and fused GroupNorm sums checked against the same output relabelled on the host."""
    from octfusion_amd import _lib, modules as M, ops
    B, d, cin, cout, mode = 6, 6, 128, 128, 3
    oc, doc, _, _ = shell6(2)
    saved = ops.get_precision()
    ops.set_precision('fp16x3')
    try:
        conv = M.GraphConv(cin, cout, 7, 7, d - 1)
        gn = M.DualOctreeGroupNorm(cin)
        sd = C.fill_state_dict([('c.' + k, tuple(v.shape)) for k, v in conv.state_dict().items()] +
                               [('g.' + k, tuple(v.shape)) for k, v in gn.state_dict().items()])
        conv.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith('c.')})
        gn.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith('g.')})
        conv, gn = conv.to(dev()), gn.to(dev())
        N = doc.csr(d)[2]
        x = C.rand_input('lab_x', N, cin).to(dev())
        emb = C.rand_input('lab_e', B, cout).to(dev())
        res = C.rand_input('lab_r', N, cout).to(dev())
        # labels 0..5 in ascending runs: a 10-row element inside one wave (three labels in that wave), a boundary in
        # the middle of a wave (two labels), a boundary on a wave boundary, one on a tile boundary, a 3-row tail
        cuts = [0, 1000, 1010, 5 * 256 + 64, 40 * 256, N - 3, N]
        bid = torch.zeros(N, dtype=torch.int32)
        for b in range(B):
            bid[cuts[b]:cuts[b + 1]] = b
        bid = bid.to(dev())
        hp = gn(x, doc, d, act='silu', planes=mode)
        nt = d - 1
        pw2 = conv._pw2.get(conv.weights, cin, nt, mode)
        seg_ptr, col, _, _ = doc.csr(d)
        _lib.call('ofx_set_gconv_persistent', persistent)
        base = None
        for tile in (2, 4):
            _lib.call('ofx_set_gconv2_tile', tile)
            with ops.stats_scope(dev()):
                y0 = ops.graphconv_planes(hp, mode, seg_ptr, col, doc.ext(d), pw2, cin, nt, doc.type_frac_planes(d, nt, mode),
                                          None, None, bid, res, None, stats=None).clone()      # no labels involved
                stats = ops.stats_zeros(B * cout * 2, dev())
                y = ops.graphconv_planes(hp, mode, seg_ptr, col, doc.ext(d), pw2, cin, nt, doc.type_frac_planes(d, nt, mode),
                                         None, emb, bid, res, None, stats=stats).clone()
                st = stats.clone()
            torch.cuda.synchronize()
            want = y0 + emb[bid.long()]
            assert float((y - want).abs().max()) <= 1e-6 * float(want.abs().max())
            ws = torch.zeros(B, cout, 2, dtype=torch.float64, device=dev())
            ws[:, :, 0].index_add_(0, bid.long(), y.double())
            ws[:, :, 1].index_add_(0, bid.long(), y.double() ** 2)
            assert float((st.view(B, cout, 2) - ws).abs().max()) <= 1e-5 * float(ws.abs().max())
            if base is not None and not persistent:
                assert torch.equal(y, base)
            base = y
            assert not ops.sync_error(dev())
    finally:
        _lib.call('ofx_set_gconv_persistent', 1)
        _lib.call('ofx_set_gconv2_tile', 0)
        ops.set_precision(saved)
