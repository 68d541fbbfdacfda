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


# ---------------------------------------------------------------------------------------------------------------
# Dense lr U-Net (graph_unet_lr.UNet3DModel) in node-row layout: forward with saved intermediates + backward.
# Parameter gradients are returned under the module's state_dict key names.
def _silu_grad(x):
    s = torch.sigmoid(x)
    return s * (1 + x * (1 - s))


class _Grads(dict):
    def add(self, name, g):
        self[name] = self[name] + g if name in self else g


def _gn32_bwd(norm, x, dy, gs, act, G, prefix):
    dx, dg, db = ops.group_norm_backward(x, dy, gs.cache.batch_id(gs.depth), gs.cache.count(gs.depth), gs.B, norm.weight,
                                         norm.bias, norm.num_groups, norm.eps, act=act, count_eps=0.0)
    G.add(prefix + 'weight', dg)
    G.add(prefix + 'bias', db)
    return dx


def _conv_bwd(conv, x, dy, gs, G, prefix, need_dx=True):
    dx, dw, db = gridconv_backward(conv, x, dy, gs, need_dx=need_dx)
    G.add(prefix + 'weight', dw)
    G.add(prefix + 'bias', db)
    return dx


def _point_bwd(m, x, dy, G, prefix):
    w2 = m.weight.view(m.weight.shape[0], m.weight.shape[1])
    dx, dw, db = ops.linear_backward(x, dy, w2)
    G.add(prefix + 'weight', dw.view_as(m.weight))
    G.add(prefix + 'bias', db)
    return dx


def _linear_bwd(m, x, dy, G, prefix):
    dx, dw, db = ops.linear_backward(x, dy, m.weight)
    G.add(prefix + 'weight', dw)
    if m.bias is not None:
        G.add(prefix + 'bias', db)
    return dx


def _resnet_fwd(blk, x, emb_act, gs):
    h1 = blk.block1[0](x, gs, act='silu')
    t = blk.time_mlp[1](emb_act)
    c1 = blk.block1[2](h1, gs, emb=t)
    h2 = blk.block2[0](c1, gs, act='silu')
    skip = x if isinstance(blk.res_conv, nn.Identity) else blk.res_conv(x)
    y = blk.block2[3](h2, gs, res=skip)
    return y, (x, h1, c1, h2, gs)


def _resnet_bwd(blk, saved, emb_act, dy, G, prefix):
    """returns (dx, d emb_act contribution)."""
    x, h1, c1, h2, gs = saved
    dh2 = _conv_bwd(blk.block2[3], h2, dy, gs, G, prefix + 'block2.3.')
    dc1 = _gn32_bwd(blk.block2[0], c1, dh2, gs, 'silu', G, prefix + 'block2.0.')
    dt = batch_sums(dc1, gs.cache.batch_id(gs.depth), gs.B)
    demb = _linear_bwd(blk.time_mlp[1], emb_act, dt, G, prefix + 'time_mlp.1.')
    dh1 = _conv_bwd(blk.block1[2], h1, dc1, gs, G, prefix + 'block1.2.')
    dx = _gn32_bwd(blk.block1[0], x, dh1, gs, 'silu', G, prefix + 'block1.0.')
    if isinstance(blk.res_conv, nn.Identity):
        dx += dy
    else:
        dx += _point_bwd(blk.res_conv, x, dy, G, prefix + 'res_conv.')
    return dx, demb


def _attnseq_fwd(seq, x, gs):
    a = seq[0](x, gs, act='silu')
    blk = seq[2]
    n = blk.norm(a, gs)
    qkv = blk.qkv(n)
    h = ops.attention(qkv, gs.B, 8 ** gs.depth, blk.num_heads)
    y = blk.proj_out(h, res=a)
    return y, (x, a, n, qkv, h, gs)


def _attnseq_bwd(seq, saved, dy, G, prefix):
    x, a, n, qkv, h, gs = saved
    blk = seq[2]
    dh = _point_bwd(blk.proj_out, h, dy, G, prefix + '2.proj_out.')
    dqkv = ops.attention_backward(qkv, dh, gs.B, 8 ** gs.depth, blk.num_heads)
    dn = _point_bwd(blk.qkv, n, dqkv, G, prefix + '2.qkv.')
    da = _gn32_bwd(blk.norm, a, dn, gs, None, G, prefix + '2.norm.')
    da += dy
    return _gn32_bwd(seq[0], x, da, gs, 'silu', G, prefix + '0.')


@torch.no_grad()
def lr_unet_forward_backward(net, x_rows, batch_size, timesteps, dy_fn, label=None, as_middle=False):
    """Forward of graph_unet_lr.UNet3DModel.forward_rows keeping what the backward needs, then the backward.
    dy_fn(y) -> dL/dy (e.g. 2 (y - target) / numel for the MSE of octfusion_model_union.py:242-269).
    Returns (y, dx_rows, {state_dict key: gradient})."""
    import math
    from .graph_unet_lr import GridState, our_Identity
    gs = GridState(batch_size, net.full_depth, x_rows.device)
    G = _Grads()
    x_in = x_rows
    x = x_rows if as_middle else net.input_emb(x_rows, gs)
    # ---- time embedding (a few KB: torch elementwise + libofx linears)
    tt = timesteps.float()[:, None]
    w = net.time_pos_emb.weights
    f = tt * w[None, :] * 2 * math.pi
    pe = torch.cat((tt, f.sin(), f.cos()), dim=-1).contiguous()
    e1 = net.time_emb[0](pe)
    a1 = ops.act(e1, 'silu')
    e2 = net.time_emb[2](a1)
    if net.num_classes is not None:
        e2 = e2 + net.label_emb(label)
    emb_act = ops.act(e2, 'silu')
    demb_act = torch.zeros_like(emb_act)
    # ---- down path
    tape = []
    hs = []
    for i, (resnet, attn, down) in enumerate(net.downs):
        x, s_res = _resnet_fwd(resnet, x, emb_act, gs)
        s_att = None
        if not isinstance(attn, our_Identity):
            x, s_att = _attnseq_fwd(attn, x, gs)
        hs.append(x)
        s_down = None
        if not isinstance(down, our_Identity):
            s_down = (x, gs)
            x, gs = down(x, gs)
        tape.append(('down', i, s_res, s_att, s_down))
    x, s_m1 = _resnet_fwd(net.mid_block1, x, emb_act, gs)
    s_ma = None
    if not isinstance(net.mid_self_attn, our_Identity):
        x, s_ma = _attnseq_fwd(net.mid_self_attn, x, gs)
    x, s_m2 = _resnet_fwd(net.mid_block2, x, emb_act, gs)
    up_tape = []
    for i, (resnet, attn, up) in enumerate(net.ups):
        skip = hs.pop()
        c_left = x.shape[1]
        x = torch.cat((x, skip), dim=1)
        x, s_res = _resnet_fwd(resnet, x, emb_act, gs)
        s_att = None
        if not isinstance(attn, our_Identity):
            x, s_att = _attnseq_fwd(attn, x, gs)
        s_up = (x, gs)
        x, gs = up(x, gs)
        up_tape.append((i, s_res, s_att, s_up, c_left))
    x_end = x
    e = net.end[0](x_end, gs, act='silu')
    y = e if as_middle else net.out(e, gs)
    # ================================================================= backward
    dy = dy_fn(y)
    de = dy if as_middle else _conv_bwd(net.out, e, dy, gs, G, 'out.')
    dx = _gn32_bwd(net.end[0], x_end, de, gs, 'silu', G, 'end.0.')
    dskips = {}
    n_down = len(net.downs)
    for i, s_res, s_att, s_up, c_left in reversed(up_tape):
        resnet, attn, up = net.ups[i]
        xu, gsu = s_up
        dx = _conv_bwd(up.conv, xu, dx, gsu, G, 'ups.%d.2.conv.' % i)
        if s_att is not None:
            dx = _attnseq_bwd(attn, s_att, dx, G, 'ups.%d.1.' % i)
        dx, d = _resnet_bwd(resnet, s_res, emb_act, dx, G, 'ups.%d.0.' % i)
        demb_act += d
        dskips[n_down - 1 - i] = dx[:, c_left:].contiguous()       # ups pop the skips in reverse order
        dx = dx[:, :c_left].contiguous()
    dx, d = _resnet_bwd(net.mid_block2, s_m2, emb_act, dx, G, 'mid_block2.')
    demb_act += d
    if s_ma is not None:
        dx = _attnseq_bwd(net.mid_self_attn, s_ma, dx, G, 'mid_self_attn.')
    dx, d = _resnet_bwd(net.mid_block1, s_m1, emb_act, dx, G, 'mid_block1.')
    demb_act += d
    for kind, i, s_res, s_att, s_down in reversed(tape):
        resnet, attn, down = net.downs[i]
        if s_down is not None:
            xd, gsd = s_down
            dx = _conv_bwd(down.op, xd, dx, gsd, G, 'downs.%d.2.op.' % i)
        if i in dskips:
            dx = dx + dskips[i]
        if s_att is not None:
            dx = _attnseq_bwd(attn, s_att, dx, G, 'downs.%d.1.' % i)
        dx, d = _resnet_bwd(resnet, s_res, emb_act, dx, G, 'downs.%d.0.' % i)
        demb_act += d
    if not as_middle:
        gs0 = GridState(batch_size, net.full_depth, x_rows.device)
        dx = _conv_bwd(net.input_emb, x_in, dx, gs0, G, 'input_emb.')
    # ---- time embedding backward
    de2 = demb_act * _silu_grad(e2)
    if net.num_classes is not None:
        gl = torch.zeros_like(net.label_emb.weight)
        gl.index_add_(0, label, de2)
        G.add('label_emb.weight', gl)
    da1 = _linear_bwd(net.time_emb[2], a1, de2, G, 'time_emb.2.')
    de1 = da1 * _silu_grad(e1)
    dpe = _linear_bwd(net.time_emb[0], pe, de1, G, 'time_emb.0.')
    half = w.shape[0]
    dsin, dcos = dpe[:, 1:1 + half], dpe[:, 1 + half:]
    G.add('time_pos_emb.weights', ((dsin * f.cos() - dcos * f.sin()) * tt * 2 * math.pi).sum(0))
    return y, dx, dict(G)


# ---------------------------------------------------------------------------------------------------------------
# Sparse hr U-Net (graph_unet_hr.UNet3DModel, graph_unet_hr.py:214-281) with the nested lr net: forward + backward.
def _gconv_bwd(conv, x, dy, doctree, d, G, prefix, need_dx=True):
    dx, dw = ops.graphconv_backward(x, dy, doctree, d, conv.weights, conv.n_node_type, need_dx=need_dx)
    G.add(prefix + 'weights', dw)
    if conv.use_bias:
        G.add(prefix + 'bias', dy.sum(0))
    return dx


def _dgn_bwd(norm, x, dy, doctree, d, act, G, prefix):
    dx, dg, db = ops.group_norm_backward(x, dy, doctree.batch_id32(d), doctree.count(d), doctree.batch_size,
                                         norm.weights, norm.bias, norm.group, norm.eps, act=act)
    G.add(prefix + 'weights', dg.view(1, -1))
    G.add(prefix + 'bias', db.view(1, -1))
    return dx


# Activation checkpointing (ldm_diffusion_util.py:158-169; GraphResBlockEmbed.forward wraps _forward in
# checkpoint(..., self.use_checkpoint), modules.py:730-743; the Objaverse config sets use_checkpoint: True): a block
# built with use_checkpoint keeps only its INPUT for the backward and recomputes GroupNorm -> conv1 -> GroupNorm there
# (three of the four tensors the plain path keeps per block; at N8 = 3.25 M rows x 64..256 channels that is the
# difference between fitting a batch and not).  CHECKPOINT: None = follow the module's flag, True / False = force (A/B).
CHECKPOINT = None


def _gres_recompute(blk, x, emb_act, doctree, d):
    h1 = blk.block1_norm(x, doctree, d, act='silu')
    c1 = blk.conv1(h1, doctree, d, emb=blk.emb_layers[1](emb_act))
    h2 = blk.block2_norm(c1, doctree, d, act='silu')
    return h1, c1, h2


def _gres_fwd(blk, x, emb_act, doctree, d):
    h1, c1, h2 = _gres_recompute(blk, x, emb_act, doctree, d)
    skip = x if isinstance(blk.skip_connection, nn.Identity) else blk.skip_connection(x)
    y = blk.conv2(h2, doctree, d, res=skip)
    ckpt = blk.use_checkpoint if CHECKPOINT is None else CHECKPOINT
    return y, ((x, None, None, None) if ckpt else (x, h1, c1, h2))


def _gres_bwd(blk, saved, emb_act, dy, doctree, d, G, prefix):
    x, h1, c1, h2 = saved
    if h1 is None:                       # checkpointed block: its intermediates are recomputed from the input
        h1, c1, h2 = _gres_recompute(blk, x, emb_act, doctree, d)
    dh2 = _gconv_bwd(blk.conv2, h2, dy, doctree, d, G, prefix + 'conv2.')
    dc1 = _dgn_bwd(blk.block2_norm, c1, dh2, doctree, d, 'silu', G, prefix + 'block2_norm.')
    demb_out = batch_sums(dc1, doctree.batch_id32(d), doctree.batch_size)
    demb_act = _linear_bwd(blk.emb_layers[1], emb_act, demb_out, G, prefix + 'emb_layers.1.')
    dh1 = _gconv_bwd(blk.conv1, h1, dc1, doctree, d, G, prefix + 'conv1.')
    dx = _dgn_bwd(blk.block1_norm, x, dh1, doctree, d, 'silu', G, prefix + 'block1_norm.')
    if isinstance(blk.skip_connection, nn.Identity):
        dx += dy
    else:
        dx += _linear_bwd(blk.skip_connection.linear, x, dy, G, prefix + 'skip_connection.linear.')
    return dx, demb_act


def _pool_bwd(down, x, dout, doctree, d, G, prefix):
    """Backward of modules.pool_nodes (rows of depth d -> d-1)."""
    copy_src, gemm_rows, n_out = doctree.pool_maps(d)
    C = x.shape[1]
    numd = int(doctree.nnum[d])
    dx = torch.zeros_like(x)
    ops.rows_copy(dout, dx, n_out, dmap=copy_src)                        # leaf rows were copied
    n_ne = gemm_rows.numel()
    if n_ne:
        dP = torch.empty(n_ne, C, dtype=torch.float32, device=x.device)
        ops.rows_copy(dout, dP, n_ne, smap=gemm_rows)
        tail = x[x.shape[0] - numd:].reshape(n_ne, 8 * C)
        w2 = down.weights.view(C, 8 * C)                                 # y = tail @ w2^T
        dtail, dw, _ = ops.linear_backward(tail, dP, w2)
        dx[x.shape[0] - numd:] = dtail.view(numd, C)
        G.add(prefix + 'weights', dw.view_as(down.weights))
    else:
        G.add(prefix + 'weights', torch.zeros_like(down.weights))
    return dx


def _unpool_bwd(up, x, dout, doctree, d, G, prefix):
    """Backward of modules.unpool_nodes (rows of depth d -> d+1)."""
    copy_src, a_rows, n_copy = doctree.unpool_maps(d)
    C = x.shape[1]
    dx = torch.zeros_like(x)
    ops.rows_copy(dout, dx, n_copy, dmap=copy_src)
    n_ne = a_rows.numel()
    if n_ne:
        dU = dout[n_copy:].reshape(n_ne, 8 * C)                          # y = xa @ w2, w2 [C, 8C]
        xa = torch.empty(n_ne, C, dtype=torch.float32, device=x.device)
        ops.rows_copy(x, xa, n_ne, smap=a_rows)
        w2 = up.weights.view(C, 8 * C)
        G.add(prefix + 'weights', ops.gemm_tn(xa, dU).view_as(up.weights))
        dxa = ops.gemm(dU, ops.PackedWeight().get(w2, 'nk'))             # dU @ w2^T
        ops.rows_copy(dxa, dx, n_ne, dmap=a_rows)                        # those rows receive nothing else
    else:
        G.add(prefix + 'weights', torch.zeros_like(up.weights))
    return dx


@torch.no_grad()
def hr_unet_forward_backward(net, x, doctree, unet_lr, timesteps, dy_fn, label=None, as_middle=False):
    """graph_unet_hr.UNet3DModel: forward keeping intermediates, then backward.
    unet_lr: the nested stage run as the middle -- a graph_unet_lr net (2-stage hr), another graph_unet_hr net
    (3-stage feature net, graph_unet_union.py:80-92) or None (an hr net that is itself running as_middle).
    Returns (y, dx, grads of this net, grads of the nested net) keyed by state_dict names."""
    from . import modules as M
    from .graph_unet_lr import UNet3DModel as LrNet
    G = _Grads()
    B = doctree.batch_size
    t_emb = ops.timestep_embedding(timesteps.float(), net.model_channels)
    e1 = net.time_embed[0](t_emb)
    a1 = ops.act(e1, 'silu')
    e2 = net.time_embed[2](a1)
    if net.num_classes is not None:
        e2 = e2 + net.label_emb(label)
    emb_act = ops.act(e2, 'silu')
    demb_act = torch.zeros_like(emb_act)
    d = net.input_depth
    h = x if as_middle else net.input_blocks[0](x, doctree, d)
    hs = [h]
    enc_tape = []
    for (kind, dd, _), module in zip(net._enc, net.input_blocks[1:]):
        h_in = h
        if kind == 'res':
            h, s = _gres_fwd(module, h_in, emb_act, doctree, dd)
            enc_tape.append(('res', dd, module, s))
        else:
            p = M.pool_nodes(h_in, doctree, dd, module.downsample)
            h = module.conv(p, doctree, dd - 1)
            enc_tape.append(('down', dd, module, (h_in, p)))
        hs.append(h)
    dm = net._d_mid
    box = {}

    def decoder(hh):
        """Decoder forward from `hh`, output, loss gradient, decoder backward; returns dL/dhh."""
        nonlocal demb_act
        skips = list(hs)
        dec_tape = []
        for (kind, dd, _), module in zip(net._dec, net.output_blocks):
            if kind == 'res':
                sk = skips.pop()
                c_left = hh.shape[1]
                hh, s = _gres_fwd(module, torch.cat((hh, sk), dim=1), emb_act, doctree, dd)
                dec_tape.append(('res', dd, module, s, c_left))
            else:
                h_in = hh
                u = M.unpool_nodes(h_in, doctree, dd, module.upsample)
                hh = module.conv(u, doctree, dd + 1)
                dec_tape.append(('up', dd, module, (h_in, u), 0))
        h_end = hh
        e = net.end_norm(h_end, doctree, net.input_depth, act='silu')
        y = e if as_middle else net.out(e, doctree, net.input_depth)
        box['y'] = y
        dyv = dy_fn(y)
        de = dyv if as_middle else _gconv_bwd(net.out, e, dyv, doctree, net.input_depth, G, 'out.')
        dh = _dgn_bwd(net.end_norm, h_end, de, doctree, net.input_depth, 'silu', G, 'end_norm.')
        dskip = [None] * len(hs)
        consumed = sum(1 for t in dec_tape if t[0] == 'res')
        for idx in range(len(dec_tape) - 1, -1, -1):
            kind, dd, module, s, c_left = dec_tape[idx]
            pre = 'output_blocks.%d.' % idx
            if kind == 'res':
                dcat, dea = _gres_bwd(module, s, emb_act, dh, doctree, dd, G, pre)
                demb_act += dea
                consumed -= 1                             # the j-th res block popped hs[len(hs) - 1 - j]
                dskip[len(hs) - 1 - consumed] = dcat[:, c_left:].contiguous()
                dh = dcat[:, :c_left].contiguous()
            else:
                h_in, u = s
                du = _gconv_bwd(module.conv, u, dh, doctree, dd + 1, G, pre + 'conv.')
                dh = _unpool_bwd(module.upsample, h_in, du, doctree, dd, G, pre + 'upsample.')
        box['dskip'] = dskip
        return dh

    grads_nested = {}
    if unet_lr is None:
        dh = decoder(h)
    else:
        m1, s_m1 = _gres_fwd(net.middle_block1, h, emb_act, doctree, dm)

        def after_nested(h_lr):
            nonlocal demb_act
            hc = torch.cat((m1, h_lr), dim=1)
            hh, s_m2 = _gres_fwd(net.middle_block2, hc, emb_act, doctree, dm)
            dhh = decoder(hh)
            dhc, dea = _gres_bwd(net.middle_block2, s_m2, emb_act, dhh, doctree, dm, G, 'middle_block2.')
            demb_act += dea
            c1 = m1.shape[1]
            box['dm1_left'] = dhc[:, :c1].contiguous()
            return dhc[:, c1:].contiguous()

        if isinstance(unet_lr, LrNet):
            _, dm1_n, grads_nested = lr_unet_forward_backward(unet_lr, m1, B, timesteps, after_nested, label=label,
                                                             as_middle=True)
        else:
            _, dm1_n, grads_nested, _ = hr_unet_forward_backward(unet_lr, m1, doctree, None, timesteps, after_nested,
                                                                 label=label, as_middle=True)
        dh, dea = _gres_bwd(net.middle_block1, s_m1, emb_act, box['dm1_left'] + dm1_n, doctree, dm, G, 'middle_block1.')
        demb_act += dea
    dskip = box['dskip']
    # ---------------------------------------------------------------- backward (encoder side)
    for k in range(len(enc_tape) - 1, -1, -1):
        kind, dd, module, s = enc_tape[k]
        pre = 'input_blocks.%d.' % (k + 1)
        if dskip[k + 1] is not None:
            dh = dh + dskip[k + 1]
        if kind == 'res':
            dh, dea = _gres_bwd(module, s, emb_act, dh, doctree, dd, G, pre)
            demb_act += dea
        else:
            h_in, p = s
            dp = _gconv_bwd(module.conv, p, dh, doctree, dd - 1, G, pre + 'conv.')
            dh = _pool_bwd(module.downsample, h_in, dp, doctree, dd, G, pre + 'downsample.')
    if dskip[0] is not None:
        dh = dh + dskip[0]
    dx = dh if as_middle else _gconv_bwd(net.input_blocks[0], x, dh, doctree, net.input_depth, G, 'input_blocks.0.')
    # ---- time embedding
    de2 = demb_act * _silu_grad(e2)
    if net.num_classes is not None:
        gl = torch.zeros_like(net.label_emb.weight)
        gl.index_add_(0, label, de2)
        G.add('label_emb.weight', gl)
    da1 = _linear_bwd(net.time_embed[2], a1, de2, G, 'time_embed.2.')
    _linear_bwd(net.time_embed[0], t_emb, da1 * _silu_grad(e1), G, 'time_embed.0.')
    return box['y'], dx, dict(G), grads_nested
