"""DualOctreeGroupNorm writing operand planes + the consuming GraphConv's aux rows on the sibling-octet mapping
(`ofx_gn_apply_planes_oct`, csrc/ofx_norm.hip; reference models/networks/modules.py:291-326 feeding :194-220).

  * main rows: bit-equal to the row-strided launch it replaces (same arithmetic per element) and within fp32 rounding of
    the oracle's float64 GroupNorm + SiLU;
  * aux rows: aux[0] = zeros, aux[1 + v] = the mean over multi-neighbour segment v of the rows AS STORED (hi + lo), i.e.
    what the stand-alone pre-pass of the planes GraphConv would compute -- checked against a float64 mean of the merged
    planes over the CSR, for the rows owned by an octet AND the leftovers, and against the two older launches;
  * ragged batch with an empty element (leaf prefix not a multiple of eight: shift != 0, octets that straddle batch
    elements and the start / end of the tensor), every width class of the thread mapping (C / 4 lanes per row: 16, 24,
    32, 48, 96, 128 -> 16 ... 2 octets per block, incl. widths that leave lanes idle), the three operand formats.
"""
import pytest
import torch

import common as C
from test_gpu_fullwidth import dev, errors

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _tree(kind):
    from octfusion_amd.dual_octree import DualOctree
    from octfusion_amd.octree import split2octree_small
    from oracle import dual_octree as OD, sampler as OS
    if kind == 'ragged':
        split = C.random_split_small(5, 3, 17, p=0.45)
        split[1] = -1.0                                   # an element with nothing below the full layer
        fd, depth = 3, 5
    else:
        split = C.shell6_split(2, jitter=True)
        fd, depth = 4, 6
    doc = DualOctree(split2octree_small(split.to(dev()), depth, fd))
    o_doc = OD.OracleDualOctree(OS.split2octree_small(split, depth, fd))
    o_doc.post_processing_for_docnn()
    return doc, o_doc


def _aux_values(y, V, Cc, mode):
    from octfusion_amd import ops
    a = getattr(y, ops.AUX_ATTR)
    if ops.planes_pairs(mode):
        a = a.view(torch.float32).view(V + 1, -1)[:, :Cc]
    else:
        a = a.view(torch.float16).view(V + 1, -1)[:, :Cc]
    return ops.planes_merge(a, mode)


@pytest.mark.parametrize('kind,d,Cc,mode', [('ragged', 5, 128, 3), ('ragged', 5, 64, 3), ('ragged', 4, 96, 3),
                                            ('ragged', 5, 192, 2), ('ragged', 4, 384, 3), ('ragged', 5, 512, 3),
                                            ('ragged', 5, 128, 1), ('shell', 6, 128, 3), ('shell', 5, 256, 3),
                                            ('shell', 6, 64, 2)])
def test_octet_launch_matches_the_older_launches_and_the_oracle(kind, d, Cc, mode):
    from octfusion_amd import modules as M, ops
    from oracle import modules as OM
    doc, o_doc = _tree(kind)
    N = doc.csr(d)[2]
    B = doc.batch_size
    gn = M.DualOctreeGroupNorm(Cc)
    sd = C.fill_state_dict([(k, tuple(v.shape)) for k, v in gn.state_dict().items()])
    gn.load_state_dict(sd)
    gn = gn.to(dev())
    x = C.rand_input('gnoct_%s_%d_%d' % (kind, d, Cc), N, Cc) * 1.7 + 0.3
    ref = OM.silu(OM.dual_octree_group_norm(x.double(), o_doc, d, sd['weights'].double(), sd['bias'].double()))
    seg_ptr, col, _, _ = doc.csr(d)
    _, multi_seg, V = doc.ext(d)
    plan = doc.oct_plan(d)
    assert plan[1] == (8 - doc._leaf_base[d] % 8) % 8
    xg = x.to(dev())
    saved = (ops.AUX_PLAN, ops.GN_OCT_FINALIZE_MAX_ELEMS)
    ops.GN_OCT_FINALIZE_MAX_ELEMS = 1 << 40
    outs = {}
    try:
        for how in ('oct', 'block', '', 'oct_sepfin'):
            ops.AUX_PLAN = how.split('_')[0]
            ops.GN_OCT_FINALIZE = how != 'oct_sepfin'           # statistics finalised inside the launch (default) / by ofx_gn_finalize
            y = gn(xg, doc, d, act='silu', planes=mode)
            assert ops.planes_of(y) == mode
            outs[how] = (ops.planes_merge(y, mode), _aux_values(y, V, Cc, mode))
    finally:
        ops.AUX_PLAN, ops.GN_OCT_FINALIZE_MAX_ELEMS = saved
        ops.GN_OCT_FINALIZE = True
    main, aux = outs['oct']
    # the in-launch finalize does gn_finalize_kernel's arithmetic: same bits
    assert torch.equal(main, outs['oct_sepfin'][0]) and torch.equal(aux, outs['oct_sepfin'][1])
    tol = 2e-3 if mode == 1 else (1e-5 if mode == 2 else 2e-6)
    e = errors(main, ref)
    assert e['rel_to_max'] < tol, e
    for how in ('block', ''):
        assert torch.equal(main, outs[how][0]), 'main rows differ from the %r launch' % how
    # aux rows against a float64 mean of the stored rows over the CSR
    want = torch.zeros(V + 1, Cc, dtype=torch.float64, device=dev())
    if V:
        ms = multi_seg[:V].long()
        a, b = seg_ptr[ms].long(), seg_ptr[ms + 1].long()
        lens = b - a
        seg_of_edge = torch.repeat_interleave(torch.arange(V, device=dev()), lens)
        edge = torch.repeat_interleave(a - torch.cumsum(lens, 0) + lens, lens) + torch.arange(int(lens.sum()), device=dev())
        want[1:].index_add_(0, seg_of_edge, main.double()[col[edge].long()])
        want[1:] /= lens.double()[:, None]
    scale = float(want.abs().max()) if V else 1.0
    # leftovers average the un-split fp32 values (2^-22 / 2^-17 / 2^-11 relative to a row value for fp16 / bf16 pairs / fp16)
    atol = scale * (2e-3 if mode == 1 else (2e-5 if mode == 2 else 1e-6))
    assert bool((aux[0] == 0).all())
    assert float((aux.double() - want).abs().max()) <= atol, float((aux.double() - want).abs().max()) / scale
    own = torch.zeros(V + 1, dtype=torch.bool, device=dev())
    p, shift, n_own, n_left, (o_ptr, o_ent, o_head, o_src) = plan
    own[p[o_ent:o_ent + 2 * n_own:2].long()] = True
    assert n_own + n_left == V + 1 and not bool(own[p[o_head:o_head + 4 * n_left:4].long()].any())
    if n_own:
        # rows owned by an octet are means of the STORED values in ascending row order: exactly what the block-owned rows
        # of the round-4 launch hold wherever both launches own the row
        pb, _ = doc.aux_plan(d)
        mb = (N + 63) // 64
        own_b = torch.zeros(V + 1, dtype=torch.bool, device=dev())
        own_b[pb[mb + 1:mb + 1 + int(pb[mb])].long()] = True
        both = own & own_b
        assert bool(both.any())
        assert torch.equal(aux[both], outs['block'][1][both])
    for how in ('block', ''):
        assert float((aux - outs[how][1]).abs().max()) <= atol


def test_octet_launch_feeds_the_planes_graphconv():
    """GroupNorm (octet launch) -> planes GraphConv against the oracle in float64 on the ragged tree: the aux rows are
    consumed through the branch-free gather table, so a wrong or missing aux row shows up in the convolution."""
    from octfusion_amd import modules as M, ops
    from oracle import modules as OM
    doc, o_doc = _tree('ragged')
    d, cin, cout, nt = 5, 128, 128, 4
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, nt, use_bias=True)
    gn = M.DualOctreeGroupNorm(cin)
    sd = C.fill_state_dict([('c.' + k, tuple(v.shape)) for k, v in conv.state_dict().items()] +
                           [('g.' + k, tuple(v.shape)) for k, v in gn.state_dict().items()])
    conv.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith('c.')})
    gn.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith('g.')})
    conv, gn = conv.to(dev()), gn.to(dev())
    x = C.rand_input('gnoct_conv', N, cin)
    h_ref = OM.silu(OM.dual_octree_group_norm(x.double(), o_doc, d, sd['g.weights'].double(), sd['g.bias'].double()))
    ref = OM.graph_conv(h_ref, o_doc, d, sd['c.weights'].double(), sd['c.bias'].double(), nt)
    saved = (ops.AUX_PLAN, ops.PLANES_MIN_TILES)
    ops.PLANES_MIN_TILES = 1
    try:
        for how in ('oct', 'block'):
            ops.AUX_PLAN = how
            hp = gn(x.to(dev()), doc, d, act='silu', planes=ops.planes_mode())
            y = conv(hp, doc, d)
            e = errors(y, ref)
            assert e['rel_to_max'] < 2e-5, (how, e)
    finally:
        ops.AUX_PLAN, ops.PLANES_MIN_TILES = saved
