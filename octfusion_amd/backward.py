"""Training path: backward passes assembled from libofx's gradient kernels.

The reference trains through torch.autograd over index_select / scatter_add / mm (modules.py); here every op of
the denoising block has an explicit backward built on ofx_graphconv_bwd_data / _bwd_weight, ofx_gn_backward and
ofx_gemm_tn_f32.  Intermediates are recomputed from the block input (the reference wraps its blocks in
torch.utils.checkpoint the same way: modules.py:52-61, 743), so nothing but x, emb and dy is kept.
"""
import torch
import torch.nn as nn

from . import ops


def batch_sums(x, batch_id, batch_size):
    """[B, C] fp32: column sums per batch element (the reduction behind the `h[batch_id == i] += emb[i]` loop)."""
    x, ldx = ops._row_major(x)
    n, C = x.shape
    sums = torch.empty(batch_size * C * 2, dtype=torch.float64, device=x.device)
    ops.call('ofx_gn_stats', ops.ptr(x), ldx, n, C, ops.ptr(batch_id), batch_size, ops.ptr(sums), ops.stream())
    return sums.view(batch_size, C, 2)[:, :, 0].float()


@torch.no_grad()
def graph_resblock_embed_backward(blk, x, emb, doctree, depth, dy):
    """Gradients of y = GraphResBlockEmbed(x, emb) (modules.py:730-763) given dy.
    Returns (dx, demb, {parameter name: gradient}) with the module's own state_dict key names."""
    bid, cnt, B = doctree.batch_id32(depth), doctree.count(depth), doctree.batch_size
    n1, n2 = blk.block1_norm, blk.block2_norm
    nt = blk.conv1.n_node_type
    # ---- recompute the forward intermediates
    h1 = n1(x, doctree, depth, act='silu')
    emb_act = ops.act(emb, 'silu')
    lin = blk.emb_layers[1]
    emb_out = lin(emb_act)
    c1 = blk.conv1(h1, doctree, depth, emb=emb_out)
    h2 = n2(c1, doctree, depth, act='silu')
    grads = {}
    # ---- y = conv2(h2) + skip(x)
    dh2, grads['conv2.weights'] = ops.graphconv_backward(h2, dy, doctree, depth, blk.conv2.weights, nt)
    dc1, dg, db = ops.group_norm_backward(c1, dh2, bid, cnt, B, n2.weights, n2.bias, n2.group, n2.eps, act='silu')
    grads['block2_norm.weights'], grads['block2_norm.bias'] = dg.view(1, -1), db.view(1, -1)
    # c1 = conv1(h1) + emb_out[batch_id]
    demb_out = batch_sums(dc1, bid, B)
    demb_act, grads['emb_layers.1.weight'], grads['emb_layers.1.bias'] = ops.linear_backward(emb_act, demb_out, lin.weight)
    s = torch.sigmoid(emb)
    demb = demb_act * (s * (1 + emb * (1 - s)))                 # SiLU'(emb), [B, emb_channels]: a few KB
    dh1, grads['conv1.weights'] = ops.graphconv_backward(h1, dc1, doctree, depth, blk.conv1.weights, nt)
    dx, dg, db = ops.group_norm_backward(x, dh1, bid, cnt, B, n1.weights, n1.bias, n1.group, n1.eps, act='silu')
    grads['block1_norm.weights'], grads['block1_norm.bias'] = dg.view(1, -1), db.view(1, -1)
    if isinstance(blk.skip_connection, nn.Identity):
        dx += dy
    else:
        sl = blk.skip_connection.linear
        dxs, grads['skip_connection.linear.weight'], dbs = ops.linear_backward(x, dy, sl.weight)
        if sl.bias is not None:
            grads['skip_connection.linear.bias'] = dbs
        dx += dxs
    return dx, demb, grads


@torch.no_grad()
def gridconv_backward(conv, x, dy, gs, need_dx=True):
    """Gradients of y = GridConv3d(x) (nn.Conv3d 3^3, stride 1 / stride 2 / nearest-upsample + conv, in node-row
    layout) given dy: (dx [n_in, cin], dweight [cout, cin, 3, 3, 3], dbias [cout])."""
    from ._lib import call, ptr, stream, lib
    x, ldx = ops._row_major(x)
    dy, ldy = ops._row_major(dy)
    cin, cout, mode = conv.in_channels, conv.out_channels, conv.mode
    d_out = conv.out_depth(gs.depth)
    n_in, n_out = x.shape[0], dy.shape[0]
    assert n_out == gs.B * 8 ** d_out and dy.shape[1] == cout and x.shape[1] == cin
    dev = x.device
    ws = ops.workspace(dev)
    dx = None
    if need_dx:
        rv = gs.cache.rev(mode, d_out)
        wt = ops.PackedConv3d().get(conv.weight.detach().transpose(0, 1).contiguous())
        dx = torch.empty(n_in, cin, dtype=torch.float32, device=dev)
        fast = cout % 32 == 0 and ldy % 4 == 0
        aux = torch.empty((rv['V'] + 1) * ldy, dtype=torch.float32, device=dev) if fast else None
        call('ofx_gridconv_bwd_data', ptr(dy), ldy, cout, n_out, n_in, ptr(rv['nbr']), ptr(rv['rev_ptr']),
             ptr(rv['rev_row']), ptr(rv['rev_w']), ptr(rv['nbr_ext']) if fast else None,
             ptr(rv['multi_seg']) if fast else None, rv['V'] if fast else 0, ptr(aux), ptr(wt.t), cin, ptr(dx), cin,
             ptr(ws), ws.numel(), stream())
    Kp = lib().ofx_conv3d_packed_k(cin)
    dwp = torch.empty(Kp, cout, dtype=torch.float32, device=dev)
    fast = cin % 32 == 0 and ldx % 4 == 0
    call('ofx_gridconv_bwd_weight', ptr(x), ldx, cin, n_in, n_out, None if fast else ptr(gs.cache.table(mode, d_out, False)),
         ptr(gs.cache.table(mode, d_out, True)) if fast else None, ptr(ops.zero_row(dev)) if fast else None,
         ptr(dy), ldy, cout, ptr(dwp), ptr(ws), ws.numel(), stream())
    dweight = dwp[:27 * cin].view(27, cin, cout).permute(2, 1, 0).reshape(cout, cin, 3, 3, 3).contiguous()
    return dx, dweight, dy.sum(0)
