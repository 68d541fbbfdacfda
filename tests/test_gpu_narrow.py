"""The two gather-shaped GraphConvs of a U-Net (csrc/ofx_narrow.hip) against the oracle in float64
(reference models/networks/modules.py:194-220; call sites graph_unet_hr.py:116 and :205-209).

  * input convolution (3 / 8 channels -> 64 / 128): exact fp32 FMA -> 2e-6 rel-to-max; fused GroupNorm statistics incl.
    blocks that hold a batch boundary (ragged batch: boundaries fall inside 64-row blocks), bias, output written into a
    column slice of a wider buffer (the zero-copy skip concatenation), nt = 0;
  * output convolution (64 / 128 channels -> 3 / 8) as project-then-aggregate: the dense projection runs in the default
    contraction precision (fp16x3) -> 2e-5; multi-neighbour segments, empty segments, bias;
  * both through `modules.GraphConv.forward`, with the switches off as the A/B (the contraction kernels they replace).
"""
import pytest
import torch

import common as C
from test_gpu_fullwidth import dev, errors

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _trees():
    from octfusion_amd.dual_octree import DualOctree
    from octfusion_amd.octree import split2octree_small
    from oracle import dual_octree as OD, sampler as OS
    split = C.random_split_small(5, 3, 91, p=0.45)
    split[2] = -1.0                                   # an element with nothing below the full layer
    doc = DualOctree(split2octree_small(split.to(dev()), 5, 3))
    o_doc = OD.OracleDualOctree(OS.split2octree_small(split, 5, 3))
    o_doc.post_processing_for_docnn()
    return doc, o_doc


@pytest.mark.parametrize('cin,cout,d,nt,bias', [(3, 128, 5, 4, False), (8, 128, 5, 4, True), (3, 64, 4, 3, True),
                                                (4, 128, 5, 0, False), (8, 64, 5, 4, False), (3, 64, 5, 7, False),
                                                (4, 128, 5, 8, True)])
def test_narrow_input_conv(cin, cout, d, nt, bias):
    from octfusion_amd import modules as M, ops
    from oracle import modules as OM
    doc, o_doc = _trees()
    N = doc.csr(d)[2]
    B = doc.batch_size
    conv = M.GraphConv(cin, cout, 7, 7, nt, use_bias=bias)
    sd = C.fill_state_dict([(k, tuple(v.shape)) for k, v in conv.state_dict().items()])
    conv.load_state_dict(sd)
    conv = conv.to(dev())
    x = C.rand_input('nin_%d_%d_%d' % (cin, cout, d), N, cin)
    ref = OM.graph_conv(x.double(), o_doc, d, sd['weights'].double(), sd['bias'].double() if bias else None, nt)
    assert ops.narrow_in_ok(cin, cout, nt if nt > 1 else 0)
    wide = torch.full((N, cout + 64), 7.0, device=dev())
    ops.NARROW_IN_TAB_MIN_ROWS = 0                    # the table-driven launch at any size (product: from 512 k rows up)
    with ops.stats_scope(dev()):
        y = conv(x.to(dev()), doc, d, out=wide[:, 64:])
        st = ops.get_stats(y)
        assert y.data_ptr() == wide[:, 64:].data_ptr() and bool((wide[:, :64] == 7.0).all())
        e = errors(y, ref)
        assert e['rel_to_max'] < 2e-6, e
        if N * cout >= (1 << 14):
            assert st is not None
            bid = doc.batch_id32(d).long()
            want = torch.zeros(B, cout, 2, dtype=torch.float64, device=dev())
            want[:, :, 0].index_add_(0, bid, y.double())
            want[:, :, 1].index_add_(0, bid, y.double() ** 2)
            assert float((st.view(B, cout, 2) - want).abs().max()) <= 1e-6 * float(want.abs().max())
    # the CSR-walking launch of round 5 (one block per 64 rows) against the table-driven persistent one (the default)
    ops.NARROW_IN_TAB = False
    try:
        with ops.stats_scope(dev()):
            y_csr = conv(x.to(dev()), doc, d)
    finally:
        ops.NARROW_IN_TAB = True
        ops.NARROW_IN_TAB_MIN_ROWS = 1 << 19
    assert errors(y_csr, ref)['rel_to_max'] < 2e-6
    assert float((y_csr - y).abs().max()) <= 2e-6 * float(ref.abs().max())
    # A/B: the contraction path it replaces gives the same operator
    ops.NARROW_IN = False
    try:
        y_old = conv(x.to(dev()), doc, d)
    finally:
        ops.NARROW_IN = True
    assert errors(y_old, ref)['rel_to_max'] < 2e-4


@pytest.mark.parametrize('cin,cout,d,nt,bias', [(128, 3, 5, 4, False), (64, 3, 5, 4, True), (128, 8, 5, 4, True),
                                                (64, 4, 4, 0, False)])
def test_narrow_output_conv(cin, cout, d, nt, bias):
    from octfusion_amd import modules as M, ops
    from oracle import modules as OM
    doc, o_doc = _trees()
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, nt, use_bias=bias)
    conv.emit_stats = False
    sd = C.fill_state_dict([(k, tuple(v.shape)) for k, v in conv.state_dict().items()])
    conv.load_state_dict(sd)
    conv = conv.to(dev())
    x = C.rand_input('nout_%d_%d_%d' % (cin, cout, d), N, cin)
    ref = OM.graph_conv(x.double(), o_doc, d, sd['weights'].double(), sd['bias'].double() if bias else None, nt)
    assert N >= 4096, N
    y = conv(x.to(dev()), doc, d)
    e = errors(y, ref)
    assert e['rel_to_max'] < 2e-5, e
    ops.NARROW_OUT = False
    try:
        y_old = conv(x.to(dev()), doc, d)
    finally:
        ops.NARROW_OUT = True
    assert errors(y_old, ref)['rel_to_max'] < 2e-4
    # the aggregation alone, exact: P computed in float64 on the host
    pno = conv._pno.get(conv.weights, cin, nt if nt > 1 else 0)
    P = (x.double() @ pno.wd.double().cpu()).float().to(dev())
    seg_ptr, col, _, _ = doc.csr(d)
    ntc = nt if nt > 1 else 0
    from octfusion_amd._lib import call, ptr, stream
    out = torch.empty(N, cout, device=dev())
    tt = pno.type_term(doc.type_frac(d, ntc) if ntc else None, ntc, N, cin, conv.bias if bias else None)
    call('ofx_graphconv_narrow_out', ptr(P), P.stride(0), cout, N, ptr(seg_ptr), ptr(col), ptr(tt), ptr(out), cout, stream())
    assert errors(out, ref)['rel_to_max'] < 2e-6


@pytest.mark.parametrize('C_,planes', [(128, 0), (64, 3), (256, 3)])
def test_downsample_on_a_row_strided_input(C_, planes):
    """Downsample (modules.py:391-395) when x is a column slice of the skip-concatenation buffer (row pitch != width): the
    gather-GEMM (ofx_gather_gemm_f32, table 8 r + j) against x.view(-1, 8 C) @ W^T in float64 and against the dense GEMM on
    a contiguous copy -- with scattered output rows and with the pair-planes epilogue the pooled tensor is written in."""
    from octfusion_amd import modules as M, ops
    n = 1000
    g = torch.Generator().manual_seed(C_)
    wide = torch.randn(8 * n, C_ + 96, generator=g)
    x = wide[:, 32:32 + C_]
    down = M.Downsample(C_)
    w = C.fill_state_dict([('weights', (C_, C_, 8))])['weights']
    down.load_state_dict({'weights': w})
    down = down.to(dev())
    ref = x.double().reshape(n, 8 * C_) @ w.double().reshape(C_, 8 * C_).t()
    xg = wide.to(dev())[:, 32:32 + C_]
    assert xg.stride(0) != C_
    rows = torch.randperm(n + 50, generator=g)[:n].to(torch.int32).to(dev())
    out = torch.zeros(n + 50, C_, device=dev())
    y = down(xg, out=out, out_rows=rows, out_planes=planes)
    got = ops.planes_merge(_as_planes(out, planes)) if planes else out
    assert errors(got[rows.long()], ref)['rel_to_max'] < 2e-5
    out2 = torch.zeros(n + 50, C_, device=dev())
    down(xg.contiguous(), out=out2, out_rows=rows, out_planes=planes)
    got2 = ops.planes_merge(_as_planes(out2, planes)) if planes else out2
    assert float((got - got2).abs().max()) <= 2e-6 * float(got2.abs().max())


def _as_planes(t, mode):
    from octfusion_amd import ops
    setattr(t, ops.PLANES_ATTR, mode)
    return t
